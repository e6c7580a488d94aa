"""Flags and derived settings of the unconditional LOCO-Edit path: the subset of
reference ``src/utils/define_argparser.py:14-258``: the complete flag schema (same
names / types / defaults, so every shipped ``scripts/main_*.sh`` argument list
parses), the derived fields of ``preset`` for the unconditional branch, plus the
deployment flags ``--ckpt_path``, ``--synthetic_weights``, ``--max_batch``, ``--precision``.
"""
import argparse
import os
import random
import shutil

import numpy as np
import torch


def str2bool(v):
    """define_argparser.py:128-136."""
    if isinstance(v, bool):
        return v
    if v.lower() in ('true'):
        return True
    elif v.lower() in ('false'):
        return False
    else:
        raise argparse.ArgumentTypeError('Boolean value expected.')


# Flag schema = reference define_argparser.py:18-124 (every flag the twelve shipped scripts pass parses here with
# the reference's name, type and default); rows: (name, type, default).  Flags that only steer subsystems outside
# the unconditional hot path (prompts, guidance scales, SAM model name, ablations) are accepted and carried on the
# Namespace so the shipped scripts run unmodified; `preset` rejects the text-to-image model names.
_S, _I, _F = str, int, float
_FLAGS = [
    # default setting
    ('sh_file_name', _S, ''), ('device', _S, 'cuda:0'), ('dtype', _S, 'fp32'), ('seed', _I, 0),
    ('result_folder', _S, './runs/'), ('cache_folder', _S, ''), ('dataset_root', _S, ''),
    # model, dataset
    ('model_name', _S, 'CelebA_HQ_HF'), ('dataset_name', _S, 'Synthetic'), ('num_imgs', _I, 100),
    ('image_size', _I, 256), ('c_in', _I, 3), ('sample_idx', _I, 0),
    # prompts (text-to-image only)
    ('for_prompt', _S, ''), ('inv_prompt', _S, ''), ('neg_prompt', _S, ''),
    # diffusion schedule
    ('for_steps', _I, 100), ('inv_steps', _I, 100), ('performance_boosting_t', _F, 0.0),
    ('use_yh_custom_scheduler', 'bool', 'True'),
    # guidance (text-to-image only)
    ('guidance_scale', _F, 0), ('guidance_scale_edit', _F, 4.0),
    # edit
    ('edit_prompt', _S, ''), ('original_prompt', _S, ''), ('edit_xt', _S, 'default'),
    ('use_x_space_guidance', 'bool', 'False'), ('x_space_guidance_direct', 'bool', 'False'),
    ('x_space_guidance_edit_step', _F, 1), ('x_space_guidance_scale', _F, 0), ('x_space_guidance_num_step', _I, 0),
    ('x_space_guidance_use_edit_prompt', 'bool', 'True'), ('pca_rank_null', _I, 5), ('pca_rank', _I, 5),
    ('h_t', _F, 0.8), ('edit_t', _F, 1.0), ('no_edit_t', _F, 0.5), ('h_edit_step_size', _F, 0),
    ('x_edit_step_size', _F, 0),
    # memory (accepted for script compatibility; batches stay in HBM)
    ('pca_device', _S, 'cpu'), ('buffer_device', _S, 'cpu'), ('save_result_as', _S, 'image'),
    # experiments
    ('note', _S, None),
    ('run_cfg_forward', 'bool', 'False'), ('run_mcg_forward', 'bool', 'False'), ('run_pfg_forward', 'bool', 'False'),
    ('run_ddim_forward', 'bool', 'False'), ('run_ddim_inversion', 'bool', 'False'),
    ('run_edit_local_encoder_pullback_zt', 'bool', 'False'), ('run_edit_local_decoder_pullback_zt', 'bool', 'False'),
    ('run_edit_local_encoder_decoder_pullback_zt', 'bool', 'False'), ('encoder_decoder_by_et', 'bool', 'False'),
    ('use_mask', 'bool', 'True'), ('run_edit_local_x0_decoder_pullback_zt', 'bool', 'False'),
    ('run_edit_local_pca_zt', 'bool', 'False'), ('run_edit_null_space_projection', 'bool', 'False'),
    ('run_edit_null_space_projection_zt', 'bool', 'False'),
    ('run_edit_null_space_projection_zt_semantic', 'bool', 'False'),
    ('run_edit_null_space_projection_xt', 'bool', 'False'),
    ('run_edit_null_space_projection_xt_semantic', 'bool', 'False'),
    ('group_edit_null_space_projection', 'bool', 'False'),
    ('vis_num', _I, 4), ('choose_sem', _S, 'hair'), ('null_space_projection', 'bool', 'False'),
    # mode
    ('debug_mode', 'bool', 'False'), ('sampling_mode', 'bool', 'False'), ('non_semantic', 'bool', 'False'),
    # mask segmentation
    ('mask_model_name', _S, 'facebook/sam-vit-large'), ('filter_mask', _I, 100), ('mask_index', _I, 0),
    ('vT_path', _S, ''), ('vT1_path', _S, ''), ('jacobian', 'bool', 'False'), ('use_sega', 'bool', 'False'),
    ('edit_t_idx', _I, 1), ('num_inference_steps', _I, 3), ('random_edit', 'bool', 'False'),
]
_UNET_PRESETS = {'celeba_ddpm': 'CELEBA_DDPM', 'ffhq_p2': 'FFHQ_P2', 'tiny_ddpm': 'TINY_DDPM', 'mid_ddpm': 'MID_DDPM',
                 'tiny_adm': 'TINY_ADM', 'if64_standin': 'IF64_STANDIN', 'sd64_standin': 'SD64_STANDIN',
                 'sd64_xattn_standin': 'SD64_XATTN_STANDIN', 'tiny_latent': 'TINY_LATENT', 'tiny_latent_xattn': 'TINY_LATENT_XATTN',
                 'if64_xattn_standin': 'IF64_XATTN_STANDIN', 'tiny_adm_xattn': 'TINY_ADM_XATTN', 'sd15_unet': 'SD15_UNET', 'sd21_base_unet': 'SD21_BASE_UNET',
                 'tiny_ldm': 'TINY_LDM', 'if_i_m_unet': 'IF_I_M_UNET', 'tiny_if': 'TINY_IF', 'mid_if': 'MID_IF'}
_VAE_PRESETS = {'sd_vae_decoder': 'SD_VAE_DECODER', 'tiny_decoder': 'TINY_DECODER'}
_TILDA_V = ["proj_null[for-null](edit-null)-direct", "(for-edit)-direct", "(edit-null)-direct",
            "null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)",
            "edit-proj[for](edit)", "null+for+edit-proj[for](edit)"]


def build_parser():
    p = argparse.ArgumentParser()
    for name, typ, default in _FLAGS:
        p.add_argument('--' + name, type=str2bool if typ == 'bool' else typ, default=default)
    p.add_argument('--mask_type', type=str, default="SAM", choices=["SAM", "diffedit"])
    p.add_argument('--ablation_method', type=str, choices=["null-space-proj", "sega", "diffedit"])
    p.add_argument('--tilda_v_score_type', type=str, choices=_TILDA_V)
    # deployment flags of this build (not in the reference)
    p.add_argument('--ckpt_path', type=str, default='', help='state_dict in the vendored Ho-DDPM / ADM / diffusers key layout')
    p.add_argument('--synthetic_weights', type=int, default=None, help='seed of the deterministic weight synthesiser')
    p.add_argument('--max_batch', type=int, default=0,
                   help='largest image/probe batch resident on the GPU per pass; 0 = from the probe counts: '
                        'clamp(pca_rank + pca_rank_null, 8, 32) for the unconditional models (about 1 GB of arena per '
                        'probe at 256x256; wider batches fill the deep levels: 64 probes run 9 %% faster at 32 than at 8), '
                        '8 for the text-to-image paths (several engine contexts)')
    p.add_argument('--unet_preset', type=str, default=None, choices=sorted(_UNET_PRESETS),
                   help='override the architecture --model_name implies (small parity-test sizes)')
    p.add_argument('--vae_preset', type=str, default=None, choices=sorted(_VAE_PRESETS),
                   help='latent T-LOCO: decoder architecture (default: the Stable Diffusion autoencoder decoder geometry)')
    p.add_argument('--vae_ckpt_path', type=str, default='', help='latent T-LOCO: decoder state_dict in latent-diffusion `Decoder` naming')
    p.add_argument('--prompt_emb_path', type=str, default='',
                   help="T-LOCO: torch file {'for','edit','null': [1, tokens, D]} of prompt embeddings (the text encoder is out of scope)")
    p.add_argument('--cond_dim', type=int, default=16, help='T-LOCO stand-in: width of seeded prompt embeddings when no file is given')
    p.add_argument('--precision', type=str, default=None, choices=['f32', 'bf16x3', 'f16'],
                   help="conv arithmetic of the HIP engine: 'f32' exact fp32 MFMA (parity anchor), 'bf16x3' split-bf16 "
                        "(fp32-faithful to ~2^-16, default), 'f16' single f16 MFMA with fp32 accumulate (2^-11 operands; opt-in: "
                        "operands are cast without range management, so tangent / cotangent entries below 6e-5 lose bits and "
                        "above 65504 overflow -- pinned on synthetic weights only)")
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.max_batch <= 0:
        t2i = any(s in args.model_name for s in ('stable-diffusion', 'DeepFloyd', 'LCM'))
        # sized by what actually shares a pass: the probes of the modify-space and null-space solves when they run as a
        # pair (solver.local_basis_pair; not with LOCO_PAIR_SOLVES=0, without projection or with a loaded basis), and the
        # edited frames of all shown directions in the decode (vis_num_pc x (2 vis_num + 1) frames at most)
        pair = (args.null_space_projection and not args.vT_path and os.environ.get("LOCO_PAIR_SOLVES", "1") != "0")
        probes = args.pca_rank + (args.pca_rank_null if pair else 0)
        frames = args.pca_rank * (2 * max(int(args.vis_num), 1) + 1)     # main.py passes vis_num_pc = pca_rank
        args.max_batch = 8 if t2i else min(32, max(8, probes, frames))
    if args.unet_preset:
        from . import config
        args.unet_config = getattr(config, _UNET_PRESETS[args.unet_preset])
    if args.vae_preset:
        from . import config
        args.vae_config = getattr(config, _VAE_PRESETS[args.vae_preset])
    return args


UNCOND_MODELS_HF = ('CelebA_HQ_HF', 'LSUN_church_HF', 'LSUN_bedroom_HF', 'FFHQ_HF')
UNCOND_MODELS_P2 = ('FFHQ_P2', 'AFHQ_P2', 'Flower_P2', 'Cub_P2', 'Metface_P2')


def preset(args):
    """define_argparser.py:138-249 for the unconditional branch."""
    if args.seed == 0:
        args.seed = int(torch.randint(2**32, ()))
    seed_everything(args.seed)
    # routing by model name, define_argparser.py:147-165
    args.is_stable_diffusion = 'stable-diffusion' in args.model_name
    args.is_DeepFloyd_IF_diffusion = (not args.is_stable_diffusion) and 'DeepFloyd' in args.model_name
    args.is_LCM = (not args.is_stable_diffusion) and (not args.is_DeepFloyd_IF_diffusion) and 'LCM' in args.model_name
    if args.is_LCM:
        raise NotImplementedError('the latent-consistency path (EditLatentConsistency, edit.py:42-481: LCM scheduler, guidance '
                                  'embedding) is outside this build; Stable Diffusion is loco_edit_amd.tloco_sd, DeepFloyd-IF '
                                  'loco_edit_amd.tloco')
    if args.is_stable_diffusion:
        return _preset_t2i(args, 'Stable_Diffusion')
    if args.is_DeepFloyd_IF_diffusion:
        return _preset_t2i(args, 'DeepFloyd-IF')
    # model-name gate of define_argparser.py:166-176 (`unet_config` = an explicit architecture, tests / tiny configs)
    if getattr(args, 'unet_config', None) is None:
        if args.model_name == 'CelebA_HQ':
            raise NotImplementedError('Model weight deprecated...')
        if args.model_name in ('LSUN_bedroom', 'LSUN_cat', 'LSUN_horse'):
            raise NotImplementedError('Please download P2 weight from https://github.com/jychoi118/P2-weighting')
        if args.model_name not in UNCOND_MODELS_HF + UNCOND_MODELS_P2:
            raise ValueError('model_name choice: [CelebA_HQ_HF, LSUN_church_HF, FFHQ_HF]')
    args.exp = f'{args.model_name}-{args.dataset_name}'
    args.exp_folder = os.path.join(args.result_folder, args.exp)
    os.makedirs(args.exp_folder, exist_ok=True)
    # run-dir snapshot of the launching script (define_argparser.py:191-194), when it exists
    sh = os.path.join('scripts', args.sh_file_name)
    if args.sh_file_name and os.path.exists(sh):
        shutil.copy(sh, os.path.join(args.exp_folder, args.sh_file_name))
    args.obs_folder = os.path.join(args.exp_folder, 'obs')
    args.result_folder = os.path.join(args.exp_folder, 'results')
    os.makedirs(args.obs_folder, exist_ok=True)
    os.makedirs(args.result_folder, exist_ok=True)
    args.device = torch.device(args.device)
    args.dtype = torch.float32 if args.dtype == 'fp32' else torch.float16
    print(f'device : {args.device}, dtype : {args.dtype}')
    args.c_in = 3
    args.image_size = 256 if getattr(args, 'unet_config', None) is None else args.unet_config.resolution
    args.memory_bound = 50
    args.noise_schedule = 'linear'
    # asserts of define_argparser.py:245-247
    assert args.use_yh_custom_scheduler
    assert args.for_steps == 100
    assert args.performance_boosting_t == 0.2
    return args


def _preset_t2i(args, family):
    """define_argparser.py:147-160, 212-223, 236-239 for the text-to-image branches: Stable Diffusion (latent T-LOCO,
    c_in 4, image_size 64) and DeepFloyd-IF (pixel-space T-LOCO, c_in 3, image_size 64)."""
    args.exp = f'{family}-{args.dataset_name}-{args.note}'
    args.exp_folder = os.path.join(args.result_folder, args.exp)
    os.makedirs(args.exp_folder, exist_ok=True)
    sh = os.path.join('scripts', args.sh_file_name)
    if args.sh_file_name and os.path.exists(sh):
        shutil.copy(sh, os.path.join(args.exp_folder, args.sh_file_name))
    args.obs_folder = os.path.join(args.exp_folder, 'obs')
    args.result_folder = os.path.join(args.exp_folder, 'results')
    os.makedirs(args.obs_folder, exist_ok=True)
    os.makedirs(args.result_folder, exist_ok=True)
    args.device = torch.device(args.device)
    args.dtype = torch.float32 if args.dtype == 'fp32' else torch.float16
    print(f'device : {args.device}, dtype : {args.dtype}')
    from . import config
    if args.is_stable_diffusion:
        if getattr(args, 'unet_config', None) is None:
            # the Stable Diffusion v1.x denoiser (latent-diffusion UNetModel with SpatialTransformer blocks, 859.5 M
            # parameters; BASELINE config 4).  `--unet_preset sd64_xattn_standin` selects the round-2 stand-in
            # model names of the 2.x family (the shipped scripts: stabilityai/stable-diffusion-2-1-base) get that family's
            # widths: 1024-wide prompt states, 64-channel heads, nn.Linear proj_in / proj_out (config.SD21_BASE_UNET)
            args.unet_config = config.SD21_BASE_UNET if 'stable-diffusion-2' in args.model_name else config.SD15_UNET
        if getattr(args, 'vae_config', None) is None:
            args.vae_config = config.SD_VAE_DECODER
        if getattr(args, 'vae_encoder_config', None) is None:      # vae.encode of run_DDIMinversion (engine created on use)
            args.vae_encoder_config = config.SD_VAE_ENCODER
        if getattr(args, 'dataset', None) is None and args.dataset_name != 'Random':
            # the image datasets of the latent path are read at the autoencoder's resolution (utils.py:479-486: 512)
            from .utils import FolderDataset, SyntheticDataset
            r = args.vae_encoder_config.resolution
            args.dataset = (SyntheticDataset(r, 3) if args.dataset_name == 'Synthetic'
                            else FolderDataset(args.dataset_root, res=r, numeric=args.dataset_name != 'AFHQ'))
        args.c_in = args.unet_config.in_channels           # 4
    else:
        if getattr(args, 'unet_config', None) is None:
            # the stage-I U-Net of the id the shipped scripts name (scripts/main_T2I_DeepFloydIF_null_space_projection*.sh:4,
            # DeepFloyd/IF-I-M-v1.0); `--unet_preset if64_standin` / `if64_xattn_standin` select the round-2 / 3 stand-ins
            parts = args.model_name.split("-")               # "DeepFloyd/IF-I-M-v1.0" -> size "M" (edit.py:1204)
            size = parts[2] if len(parts) > 2 and parts[2] in config.IF_I_WIDTH else "M"
            if size == "XL" and os.environ.get("LOCO_CFG_FORK", "1") == "0":
                # 4.3 B parameters x six device layouts (24 bytes per parameter) = 103 GB per independently loaded context: three of
                # them do not fit one GPU.  With the default shared parameter store (loco_fork: the guidance branches are forks of
                # one loaded context, round 6) the flow holds the parameters once -- ~103 GB + three arenas; not run at size here
                raise SystemExit("DeepFloyd/IF-I-XL-v1.0: three independently loaded contexts (LOCO_CFG_FORK=0) of the 4.3 B-parameter "
                                 "stage-I U-Net exceed one GPU's memory; leave LOCO_CFG_FORK at its default (one shared parameter store)")
            args.unet_config = config.if_stage1_config(size)
        args.c_in = 3
    args.image_size = args.unet_config.resolution          # 64: SD latents and the IF stage-I models alike
    args.memory_bound = 5
    assert args.use_yh_custom_scheduler
    assert args.performance_boosting_t <= 0
    return args


def seed_everything(seed: int):
    """define_argparser.py:251-258."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed % (2**32))
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
