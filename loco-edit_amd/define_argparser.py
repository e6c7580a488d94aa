"""Flags and derived settings of the unconditional LOCO-Edit path: the subset of
reference ``src/utils/define_argparser.py:14-258`` the hot path reads (SURVEY.md
section 5, row "Config / flags"), same names / types / defaults, plus three
deployment flags (``--ckpt_path``, ``--synthetic_weights``, ``--max_batch``).
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


def build_parser():
    p = argparse.ArgumentParser()
    # default setting
    p.add_argument('--sh_file_name', type=str, default='', help="for logging")
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--dtype', type=str, default='fp32', help="'fp32' (the uncond scripts' setting)")
    p.add_argument('--seed', type=int, default=0, help='Random seed (0 = draw one)')
    p.add_argument('--result_folder', type=str, default='./runs/')
    p.add_argument('--dataset_root', type=str, default='')
    # model, dataset
    p.add_argument('--model_name', type=str, default='CelebA_HQ_HF')
    p.add_argument('--dataset_name', type=str, default='Synthetic')
    p.add_argument('--image_size', type=int, default=256)
    p.add_argument('--c_in', type=int, default=3)
    p.add_argument('--sample_idx', type=int, default=0)
    p.add_argument('--ckpt_path', type=str, default='', help='state_dict in the vendored Ho-DDPM key layout')
    p.add_argument('--synthetic_weights', type=int, default=None, help='seed of the deterministic weight synthesiser')
    p.add_argument('--max_batch', type=int, default=8, help='largest image/probe batch resident on the GPU')
    # diffusion schedule
    p.add_argument('--for_steps', type=int, default=100)
    p.add_argument('--inv_steps', type=int, default=100)
    p.add_argument('--performance_boosting_t', type=float, default=0.0)
    p.add_argument('--use_yh_custom_scheduler', type=str2bool, default='True')
    # edit
    p.add_argument('--edit_prompt', type=str, default='')
    p.add_argument('--use_x_space_guidance', type=str2bool, default='False')
    p.add_argument('--x_space_guidance_direct', type=str2bool, default='False')
    p.add_argument('--x_space_guidance_edit_step', type=float, default=1)
    p.add_argument('--x_space_guidance_scale', type=float, default=0)
    p.add_argument('--x_space_guidance_num_step', type=int, default=0)
    p.add_argument('--pca_rank_null', type=int, default=5)
    p.add_argument('--pca_rank', type=int, default=5)
    p.add_argument('--edit_t', type=float, default=1.0)
    # memory (accepted for script compatibility; batches stay in HBM)
    p.add_argument('--pca_device', type=str, default='cpu')
    p.add_argument('--buffer_device', type=str, default='cpu')
    p.add_argument('--save_result_as', type=str, default='image')
    # experiments
    p.add_argument('--note', type=str)
    p.add_argument('--run_ddim_forward', type=str2bool, default='False')
    p.add_argument('--run_ddim_inversion', type=str2bool, default='False')
    p.add_argument('--encoder_decoder_by_et', type=str2bool, default='False')
    p.add_argument('--use_mask', type=str2bool, default='True')
    p.add_argument('--run_edit_null_space_projection', type=str2bool, default='False')
    p.add_argument('--group_edit_null_space_projection', type=str2bool, default='False')
    p.add_argument('--vis_num', type=int, default=4)
    p.add_argument('--choose_sem', type=str, default='hair')
    p.add_argument('--null_space_projection', type=str2bool, default='False')
    p.add_argument('--sampling_mode', type=str2bool, default='False')
    p.add_argument('--mask_index', type=int, default=0)
    p.add_argument('--mask_type', type=str, default="SAM", choices=["SAM", "diffedit"])
    p.add_argument('--vT_path', type=str, default="")
    p.add_argument('--vT1_path', type=str, default="")
    p.add_argument('--random_edit', type=str2bool, default='False')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def preset(args):
    """define_argparser.py:138-249 for the unconditional branch."""
    if args.seed == 0:
        args.seed = int(torch.randint(2**32, ()))
    seed_everything(args.seed)
    if any(s in args.model_name for s in ('stable-diffusion', 'DeepFloyd', 'LCM')):
        raise NotImplementedError('text-to-image T-LOCO (SD / DeepFloyd-IF / LCM) is outside this build (SURVEY.md 8f.2)')
    args.is_stable_diffusion = args.is_DeepFloyd_IF_diffusion = args.is_LCM = False
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
    args.memory_bound = 50
    args.noise_schedule = 'linear'
    # asserts of define_argparser.py:245-247
    assert args.use_yh_custom_scheduler
    assert args.for_steps == 100
    assert args.performance_boosting_t == 0.2
    return args


def seed_everything(seed: int):
    """define_argparser.py:251-258."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed % (2**32))
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
