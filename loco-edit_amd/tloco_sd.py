"""Text-supervised T-LOCO in LATENT space (Stable Diffusion path) on the MI355X engine: the class
``EditStableDiffusion`` with the method names, argument order and file names of the reference
(``src/modules/edit.py:483-1196``).

What differs from the pixel-space class (``tloco.EditDeepFloydIF``, which this one extends): the edited variable is the
latent ``z_t`` [1,4,64,64] while the Jacobian is taken of the DECODED image,

    x0_hat(z_t) = vae.decode( (z_t - sqrt(1-a_t) eps_cfg(z_t, t)) / sqrt(a_t) / 0.18215 )        (edit.py:757-781)

so J = J_dec(z0_hat) . s (I - sigma sum_c w_c J_eps,c) with s = 1 / (0.18215 sqrt(a_t)), sigma = sqrt(1 - a_t), and the
mask selects pixels of the decoded image (3 x 512 x 512).  A probe batch runs one tangent pass per CFG branch (denoiser
engines, as in the pixel-space class), one ``loco_lincomb``, and one tangent pass of the DECODER engine (arch "dec",
its own ``loco_ctx``, mask on its output); the cotangent runs the same chain backwards.  The decoder's primal is
evaluated once per solve at z0_hat.

Built: ``_classifer_free_guidance`` (:636-674), ``DDIMforwardsteps`` (:677-754, decode + PNG at the end), ``get_x0``
(:757-781), ``get_delta_zt_via_grad`` (:784-828), ``local_encoder_decoder_pullback_zt`` (:830-915),
``run_edit_null_space_projection_zt`` (:918-1041), ``run_edit_null_space_projection_zt_semantic`` (:1045-1174, also
``use_sega``), ``x_space_guidance_direct`` (:1177-1184), and ``run_DDIMinversion`` (:568-633): ``vae.encode`` on an ENCODER
engine (arch "enc", its own ``loco_ctx``, created on first use), the posterior sample (``loco_latent_sample``) and the
ascending DDIM loop on the inversion prompt.  The edit runs themselves start from z_T ~ N(0, I) (``dataset_name 'Random'``,
edit.py:937-938, as all shipped Stable Diffusion scripts do).

NOT built (stated, not hidden): the networks themselves are diffusers' ``UNet2DConditionModel`` (CLIP cross-attention)
and ``AutoencoderKL`` (un-vendored, hub weights).  The denoiser here is the guided-diffusion U-Net of the engine on 4
latent channels with a text cross-attention stage behind every attention block (``config.SD64_XATTN_STANDIN``: the
prompt's 77 x 768 encoder states go to ``loco_set_context``, one context per CFG branch; ``SD64_STANDIN`` feeds the pooled
prompt through the time embedding instead); the decoder is the
latent-diffusion ``Decoder`` module tree behind the autoencoder's ``post_quant_conv`` (``config.SD_VAE_DECODER``: the
published geometry of the SD autoencoder's decoder, 49.5 M parameters) -- a decoder state_dict in LDM naming loads
unchanged and a diffusers ``AutoencoderKL`` state_dict through ``checkpoints.hf_autoencoder_kl_to_decoder`` (key map
written from the published layout, unpinned); the denoiser stays a stand-in.
Architecture parity is therefore unpinned; the orchestration is pinned against the reference's own methods run on the
same stand-ins (oracle/make_golden_tloco_sd.py).  SAM masks are an input (``mask/mask.pt``).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

from . import solver
from .config import UNetConfig, synth_params
from .hip import LocoEngine
from .tloco import BranchStreams, EditDeepFloydIF, IFScheduler, cfg_weights
from .utils import save_image as _save_image

LATENT_SCALE = 0.18215          # edit.py:605, 748, 769


class SDScheduler(IFScheduler):
    """The scheduler the reference patches onto the Stable Diffusion pipeline (utils.py:147-157 with the
    ``set_timesteps`` / ``step`` of :172-213): the pipeline's own alpha-bar table -- ``scaled_linear`` betas
    0.00085 .. 0.012 over 1000 steps, float32, as diffusers' DDIMScheduler builds it for every SD v1/v2 checkpoint --
    float timesteps ``linspace(0,1,N)*999``, alpha-bar looked up at ``floor(t)``."""
    t_max = 999

    def __init__(self, engine=None):
        self.betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.timesteps = self.timesteps_next = None
        self.engine = engine


class LatentCFGJacobianOperator:
    """J and J^T of the decoded x0_hat(z_t) under classifier-free guidance: per-prompt denoiser engines + decoder engine."""

    def __init__(self, branches: Dict[str, LocoEngine], weights, decoder: LocoEngine, z, t, at, z0_scaled, mask, streams=None):
        self.streams = streams or BranchStreams(1, "cpu")
        self.w = [(branches[name], w) for name, w in weights if w != 0.0]
        self.lead = self.w[0][0]
        self.dec = decoder
        self.n = self.lead.n
        self.n_out = decoder.n_out      # rows of J V live on the decoded image
        self.masked = mask is not None
        a32 = np.float32(at)
        self.s = float(np.float32(1.0) / (np.float32(LATENT_SCALE) * np.sqrt(a32)))
        self.sigma = float(np.sqrt(np.float32(1.0) - a32))
        zc = z.contiguous()
        self.streams.run([(lambda e=eng: e.pmp_primal(zc, float(t), at, None, use_et=True)) for eng, _ in self.w])   # dEps products in latent space
        decoder.pmp_primal(z0_scaled.contiguous(), 0.0, 1.0, mask, use_et=True)   # raw decoder Jacobian, mask on the image

    def check_mask(self):
        if self.masked and self.dec.mask_count() == 0:
            raise ValueError("empty mask: J = d x0_hat[mask] / d z_t has no rows")

    def jvp(self, V):              # [k, n_z] -> dense masked [k, n_image]
        bufs = [torch.empty(V.shape[0], self.lead.n_out, device=V.device, dtype=torch.float32) for _ in self.w]
        outs = self.streams.run([(lambda e=eng, o=o: e.pmp_jvp(V, out=o)) for (eng, _), o in zip(self.w, bufs)])
        terms = [(self.s, V)] + [(-self.s * self.sigma * w, o) for (_, w), o in zip(self.w, outs)]
        return self.dec.pmp_jvp(self.lead.lincomb(terms))

    def vjp(self, U):              # dense [k, n_image] -> [k, n_z]
        g = self.dec.pmp_vjp(U)
        bufs = [torch.empty(g.shape[0], self.n, device=g.device, dtype=torch.float32) for _ in self.w]
        outs = self.streams.run([(lambda e=eng, o=o: e.pmp_vjp(g, out=o)) for (eng, _), o in zip(self.w, bufs)])
        terms = [(self.s, g)] + [(-self.s * self.sigma * w, o) for (_, w), o in zip(self.w, outs)]
        return self.lead.lincomb(terms)

    def gather(self, U):
        return self.dec.mask_gather(U)


class EditStableDiffusion(EditDeepFloydIF):
    def __init__(self, args):
        super().__init__(args)
        # ---- the decoder network (vae.decode, edit.py:498): its own engine context
        vcfg: UNetConfig = args.vae_config
        if vcfg.arch != "dec" or vcfg.in_channels != self.cfg.in_channels:
            raise ValueError("vae_config must be a decoder (arch 'dec') over the denoiser's latent channels")
        vparams = getattr(args, "vae_params", None)
        if vparams is None:
            if getattr(args, "vae_ckpt_path", ""):
                vparams = torch.load(args.vae_ckpt_path, map_location="cpu")
                vparams = vparams.get("state_dict", vparams)
                from .checkpoints import hf_autoencoder_kl_to_decoder, is_hf_autoencoder_kl
                if is_hf_autoencoder_kl(vparams):          # diffusers AutoencoderKL file: keep post_quant_conv + decoder.*
                    vparams = hf_autoencoder_kl_to_decoder(vparams, vcfg)
            else:
                seed = getattr(args, "synthetic_weights", None)
                if seed is None:
                    raise ValueError("no decoder checkpoint: pass --vae_ckpt_path or --synthetic_weights SEED")
                vparams = synth_params(vcfg, seed=int(seed))
        self.vae_cfg = vcfg
        self.vae_engine = LocoEngine(vcfg, max_batch=getattr(args, "max_batch", 8), device=self.device)
        self.vae_engine.load_state_dict(vparams)
        prec = getattr(args, "precision", None) or os.environ.get("LOCO_PRECISION")
        if prec:
            self.vae_engine.set_precision(prec)
        self.scheduler = SDScheduler(engine=self.engine)
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        self.edit_t_idx = int((self.scheduler.timesteps - self.edit_t * 1000).abs().argmin())
        self.use_sega = getattr(args, "use_sega", False)
        print(f'decoder : {vcfg.in_channels}x{vcfg.resolution}^2 -> {vcfg.out_ch}x{vcfg.out_resolution}^2')
        # ---- the inversion's inputs (edit.py:509, 524-529): steps, prompt embedding, image dataset; the encoder engine is
        # created on first use (run_DDIMinversion is the only caller)
        self.inv_steps = getattr(args, "inv_steps", 100)
        self.inv_prompt = getattr(args, "inv_prompt", "")
        pe = getattr(args, "prompt_emb", None)
        self.inv_prompt_emb = pe["inv"] if (pe is not None and "inv" in pe) else self.for_prompt_emb
        self.enc_engine: Optional[LocoEngine] = None
        self.dataset = getattr(args, "dataset", None)

    # ------------------------------------------------------------------ encode (edit.py:594-597)
    def _encoder(self) -> LocoEngine:
        if self.enc_engine is not None:
            return self.enc_engine
        args = self.args
        ecfg: UNetConfig = getattr(args, "vae_encoder_config", None)
        if ecfg is None:
            raise ValueError("run_DDIMinversion needs the autoencoder's encoder: set vae_encoder_config (config.SD_VAE_ENCODER)")
        if ecfg.arch != "enc" or ecfg.out_ch != 2 * self.cfg.in_channels or ecfg.out_resolution != self.cfg.resolution:
            raise ValueError("vae_encoder_config must be an encoder (arch 'enc') onto the denoiser's latent (2 z moment channels)")
        eparams = getattr(args, "vae_encoder_params", None)
        if eparams is None:
            if getattr(args, "vae_ckpt_path", ""):
                eparams = torch.load(args.vae_ckpt_path, map_location="cpu")
                eparams = eparams.get("state_dict", eparams)
                from .checkpoints import hf_autoencoder_kl_to_encoder, is_hf_autoencoder_kl
                if is_hf_autoencoder_kl(eparams):          # diffusers AutoencoderKL file: keep quant_conv + encoder.*
                    eparams = hf_autoencoder_kl_to_encoder(eparams, ecfg)
            else:
                seed = getattr(args, "synthetic_weights", None)
                if seed is None:
                    raise ValueError("no encoder checkpoint: pass --vae_ckpt_path or --synthetic_weights SEED")
                eparams = synth_params(ecfg, seed=int(seed))
        self.enc_cfg = ecfg
        self.enc_engine = LocoEngine(ecfg, max_batch=1, device=self.device)
        self.enc_engine.load_state_dict(eparams)
        prec = getattr(args, "precision", None) or os.environ.get("LOCO_PRECISION")
        if prec:
            self.enc_engine.set_precision(prec)
        print(f'encoder : {ecfg.in_channels}x{ecfg.resolution}^2 -> 2x{ecfg.out_ch // 2}x{ecfg.out_resolution}^2 moments')
        return self.enc_engine

    def encode(self, x0: torch.Tensor, noise: Optional[torch.Tensor] = None, sample: bool = True) -> torch.Tensor:
        """``self.vae.encode(x0).latent_dist.sample() * 0.18215`` (edit.py:594-597) for images [B, 3, R, R] in [-1, 1].
        ``noise``: the posterior's normal draw (default: torch.randn on the device, as diffusers draws it);
        ``sample=False``: the posterior mean."""
        enc = self._encoder()
        x = x0.to(self.device, torch.float32).contiguous()
        out = []
        for b in range(x.shape[0]):
            mom = enc.unet_forward(x[b:b + 1].contiguous(), 0.0)
            nz = None
            if sample:
                shape = (1, mom.shape[1] // 2) + tuple(mom.shape[2:])
                nz = (torch.randn(shape, device=self.device, dtype=torch.float32) if noise is None
                      else noise[b:b + 1].to(self.device, torch.float32).contiguous())
            out.append(enc.latent_sample(mom, nz, float(np.float32(LATENT_SCALE))))
        return torch.cat(out)

    @torch.no_grad()
    def run_DDIMinversion(self, idx, guidance=None, vis_traj=False, noise=None):
        """edit.py:568-633.  Prompt: (CFG) pos inv_prompt / neg null_prompt, (no CFG) inv_prompt alone; CFG only when
        ``guidance`` is given and guidance_scale > 1.  Ascending custom-scheduler timesteps, the last one is not stepped."""
        print('start DDIMinversion')
        self.EXP_NAME = f'DDIMinversion-{self.dataset_name}-{idx}-for_{self.for_prompt}-inv_{self.inv_prompt}'
        do_cfg = (self.guidance_scale > 1.0) and (guidance is not None)
        if not self.use_yh_custom_scheduler:
            raise ValueError('recommend to use yh custom scheduler')
        self.scheduler.set_timesteps(self.inv_steps, device=self.device, is_inversion=True)
        timesteps = self.scheduler.timesteps
        if self.dataset is None:
            raise ValueError("run_DDIMinversion needs an image dataset (dataset_name 'Random' has none)")
        x0 = self.dataset[idx]
        if self.sharder.is_main:
            _save_image((x0 / 2 + 0.5).clamp(0, 1), os.path.join(self.result_folder, 'original_x0.png'))
        latents = self.encode(x0, noise=noise)
        F, E, N = self.inv_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb      # the "for" branch carries inv_prompt
        for i, t in enumerate(timesteps):
            if i == len(timesteps) - 1:
                break
            noise_pred = self._classifer_free_guidance(latents, t, F, E, N, mode="null+(for-null)",
                                                       do_classifier_free_guidance=do_cfg)
            latents = self.scheduler.step(noise_pred, t, latents, eta=0).prev_sample
        return latents

    # ------------------------------------------------------------------ decode (edit.py:748-750, 769-771)
    def decode(self, z_scaled: torch.Tensor) -> torch.Tensor:
        """``self.vae.decode(z).sample`` for z already divided by 0.18215."""
        z = z_scaled.to(self.device, torch.float32).contiguous()
        mb = self.vae_engine.max_batch
        return torch.cat([self.vae_engine.unet_forward(z[b0:b0 + mb].contiguous(), 0.0) for b0 in range(0, z.shape[0], mb)])

    def _z0_scaled(self, zt, t, noise_pred):
        at = self.scheduler.alpha_at(t)
        _, z0 = self.engine.sched_step(zt, noise_pred, at, at, 0.0, None, want_x0=True)   # (z - sqrt(1-a) eps) / sqrt(a)
        return self.engine.lincomb([(float(np.float32(1.0) / np.float32(LATENT_SCALE)), z0)])

    # ------------------------------------------------------------------ x0 (edit.py:757-781)
    def get_x0(self, zt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=None,
               mode="null+(for-null)+(edit-null)", flatten=False):
        assert mode in ["null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)"]
        do_cfg = self.guidance_scale > 1.0
        zt = zt.to(self.device, torch.float32).contiguous()
        noise_pred = self._classifer_free_guidance(zt, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mode=mode,
                                                   do_classifier_free_guidance=do_cfg)
        x0_hat = self.decode(self._z0_scaled(zt, t, noise_pred))
        if mask is not None:
            return x0_hat[:, mask.to(x0_hat.device)]
        if flatten:
            x0_hat = x0_hat.view(x0_hat.shape[0], -1)
        return x0_hat

    def _operator(self, zt, t, mask, mode):
        weights = cfg_weights(mode, self.guidance_scale, self.guidance_scale_edit, self.guidance_scale > 1.0)
        zt = zt.to(self.device, torch.float32).contiguous()
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        noise_pred = self._classifer_free_guidance(zt, t, F, E, N, mode=mode, do_classifier_free_guidance=self.guidance_scale > 1.0)
        z0s = self._z0_scaled(zt, t, noise_pred)
        return LatentCFGJacobianOperator(self.branches, weights, self.vae_engine, zt, t, self.scheduler.alpha_at(t), z0s, mask,
                                         streams=self.branch_streams)

    # ------------------------------------------------------------------ solver (edit.py:830-915)
    def local_encoder_decoder_pullback_zt(self, zt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, op=None,
                                          block_idx=None, pca_rank=50, chunk_size=25, min_iter=10, max_iter=100,
                                          convergence_threshold=1e-3, mask=None, mode="null+(for-null)+(edit-null)",
                                          v0=None, verbose=True):
        assert mode in ["null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)"]
        self._bind_all(for_prompt_emb, edit_prompt_emb, null_prompt_emb)
        n = self.engine.n
        if v0 is None:
            v0 = torch.randn(n, pca_rank, device=self.device, dtype=torch.float)          # edit.py:858
        V = v0.to(self.device, torch.float32).T.contiguous()
        self.engine.qr_rows_(V)                                                            # :859
        opj = self._operator(zt, t, mask, mode)
        U, s, V, self.last_n_iter = solver.subspace_iteration(opj, self.engine, V, min_iter, max_iter,
                                                              convergence_threshold, sharder=self.sharder, verbose=verbose)
        opj.check_mask()
        u = opj.gather(U).T.contiguous()
        return u, s.sqrt(), V

    local_encoder_decoder_pullback_xt = None      # the pixel-space names do not exist on this class (edit.py:483-1196)

    # ------------------------------------------------------------------ direction through the Jacobian (edit.py:784-828)
    @torch.no_grad()
    def get_delta_zt_via_grad(self, zt, t, t_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb, mask=None,
                              mode="null+(for-null)+(edit-null)"):
        """Unit-norm J_mode^T (x0_hat[mode] - x0_hat["null+(for-null)"]) with both images decoded, restricted to the
        mask.  (Without a mask the reference reshapes the image difference with the LATENT sizes, edit.py:803-810,
        which only type-checks; here the difference keeps its image shape.)"""
        F, E, N = for_prompt_emb, edit_prompt_emb, null_prompt_emb
        x0 = self.get_x0(zt, t, t_idx, F, E, N, mask=None, mode="null+(for-null)")
        x1 = self.get_x0(zt, t, t_idx, F, E, N, mask=None, mode=mode)
        d = self.engine.lincomb([(1.0, x1.view(1, -1).contiguous()), (-1.0, x0.view(1, -1).contiguous())])
        opj = self._operator(zt, t, mask, mode)
        v_ = opj.vjp(d)                                   # the decoder's cotangent seed applies the mask
        return self.engine.null_project(v_, None)         # v_ / v_.norm(dim=1)

    get_delta_xt_via_grad = None
    get_v_modify = None

    # ------------------------------------------------------------------ sampler (edit.py:677-754)
    @torch.no_grad()
    def DDIMforwardsteps(self, zt, t_start_idx, t_end_idx, for_prompt_emb, edit_prompt_emb, null_prompt_emb,
                         mode="null+(for-null)", **kwargs):
        assert mode in ["null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)"]
        print('start DDIMforward')
        do_cfg = self.guidance_scale > 1.0
        self.scheduler.set_timesteps(self.for_steps, device=self.device)
        latents = zt.to(self.device, torch.float32).contiguous()
        for t_idx, t in enumerate(self.scheduler.timesteps):
            if t_idx < t_start_idx:
                continue
            elif t_start_idx == t_idx:
                pass
            elif t_idx == t_end_idx:
                return latents, t, t_idx
            noise_pred = self._classifer_free_guidance(latents, t, for_prompt_emb, edit_prompt_emb, null_prompt_emb,
                                                       mode=mode, do_classifier_free_guidance=do_cfg)
            latents = self.scheduler.step(noise_pred, t, latents, eta=0).prev_sample
        latents = self.engine.lincomb([(float(np.float32(1.0) / np.float32(LATENT_SCALE)), latents)])   # :748
        x0 = (self.decode(latents) / 2 + 0.5).clamp(0, 1)
        if self.sharder.is_main:
            _save_image(x0, os.path.join(self.result_folder, f'{self.EXP_NAME}.png'), nrow=x0.size(0))
        return latents, (x0 * 255).to(torch.uint8).permute(0, 2, 3, 1)

    DDPMforwardsteps = None

    @torch.no_grad()
    def run_DDIMforward(self, num_samples=5):
        """edit.py:557-566: sample `num_samples` latents from z_T ~ N(0, I) and decode them."""
        print('start DDIMforward')
        self.EXP_NAME = f'DDIMforward-for_{self.for_prompt}'
        zT = torch.randn(num_samples, self.c_in, self.image_size, self.image_size).to(device=self.device, dtype=self.dtype)
        return self.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=-1, for_prompt_emb=self.for_prompt_emb,
                                     edit_prompt_emb=self.edit_prompt_emb, null_prompt_emb=self.null_prompt_emb)

    # ------------------------------------------------------------------ drivers
    def _zT(self):
        if self.dataset_name != 'Random':
            raise ValueError("the edit runs start from z_T ~ N(0, I) (dataset_name 'Random', edit.py:937-938); "
                             "run_DDIMinversion is the image entry point")
        return torch.randn(1, self.c_in, self.image_size, self.image_size, dtype=self.dtype, device=self.device)

    def _solve_or_load(self, save_dir, zt, t, t_idx, mask, op, block_idx, pca_rank, pca_rank_null, null_space_projection,
                       modify_fn):
        """Cache protocol of edit.py:964-1000 / 1093-1123: four files decide load vs solve (rank 0 decides)."""
        os.makedirs(save_dir, exist_ok=True)
        paths = {k: os.path.join(save_dir, f) for k, f in (("um", 'u-modify.pt'), ("vm", 'vT-modify.pt'),
                 ("un", f'u-null-null_space_rank_{pca_rank_null}.pt'), ("vn", f'vT-null-null_space_rank_{pca_rank_null}.pt'))}
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        vT_null = None
        if self.sharder.agree(all(os.path.exists(p) for p in paths.values())):
            vT_modify = self._load(paths["vm"], map_location=self.device).to(self.device).type(self.dtype)
            vT_null = self._load(paths["vn"], map_location=self.device).to(self.device).type(self.dtype)
        else:
            print('subspace solve: CFG-combined Jacobian')
            u_modify, vT_modify = modify_fn()
            if self.sharder.is_main:
                if u_modify is not None:
                    torch.save(u_modify, paths["um"])
                torch.save(vT_modify, paths["vm"])
            if null_space_projection:
                u_null, s_null, vT_null = self.local_encoder_decoder_pullback_zt(
                    zt, t, t_idx, F, E, N, op=op, block_idx=block_idx, pca_rank=pca_rank_null, chunk_size=5, min_iter=10,
                    max_iter=50, convergence_threshold=1e-3, mask=~mask, mode="null+(for-null)")
                if self.sharder.is_main:
                    torch.save(u_null, paths["un"]); torch.save(vT_null, paths["vn"])
        return self.engine.null_project(vT_modify.contiguous(),
                                        vT_null[:pca_rank_null, :].contiguous() if null_space_projection else None)

    def _prepare(self, mask_index):
        self.scheduler.set_timesteps(self.for_steps)
        zT = self._zT()
        self.EXP_NAME = "original"
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        if self.sharder.agree(not os.path.exists(os.path.join(self.result_folder, "original.png"))):
            self.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E, null_prompt_emb=N,
                                  mode="null+(for-null)")                 # the image SAM would segment (edit.py:942-948)
        masks = self._masks()
        if self.sampling_mode:
            return None
        mask = masks[mask_index].squeeze(dim=0).repeat(3, 1, 1)
        zt, t, t_idx = self.DDIMforwardsteps(zT, t_start_idx=0, t_end_idx=self.edit_t_idx, for_prompt_emb=F,
                                             edit_prompt_emb=E, null_prompt_emb=N, mode="null+(for-null)")
        assert t_idx == self.edit_t_idx
        return zt, t, t_idx, mask

    @torch.no_grad()
    def run_edit_null_space_projection_zt(self, op, block_idx, vis_num, mask_index=0, vis_num_pc=1, vis_vT=False, pca_rank=50,
                                          edit_prompt=None, null_space_projection=False, pca_rank_null=50, non_semantic=False):
        """edit.py:918-1041: unsupervised directions of the decoded x0_hat, null-space projected, +/- walk, decode."""
        prep = self._prepare(mask_index)
        if prep is None:
            return None
        zt, t, t_idx, mask = prep
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        save_dir = os.path.join(self.result_folder, "basis", f'local_basis-{self.edit_t}T-pca-rank-{pca_rank}-select-mask{mask_index}')

        def modify():
            u, s, vT = self.local_encoder_decoder_pullback_zt(
                zt, t, t_idx, F, E, N, op=op, block_idx=block_idx, pca_rank=pca_rank, chunk_size=5, min_iter=10, max_iter=50,
                convergence_threshold=1e-3, mask=mask, mode="null+(for-null)")
            return u, vT
        vT = self._solve_or_load(save_dir, zt, t, t_idx, mask, op, block_idx, pca_rank, pca_rank_null, null_space_projection, modify)
        original_zt = zt.clone()
        out = None
        for pc_idx in range(vis_num_pc):
            self.EXP_NAME = (f'Edit_zt-edit_{self.edit_t}T-pc_{pc_idx}_select_mask{mask_index}_null_space_projection_'
                             f'{null_space_projection}_null_space_rank_{pca_rank_null}')
            zb = self._walk(original_zt, vT[pc_idx, :], vis_num)
        out = self.DDIMforwardsteps(zb, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                    null_prompt_emb=N, mode="null+(for-null)")      # after the loop, as in :1036-1041
        return out

    @torch.no_grad()
    def run_edit_null_space_projection_zt_semantic(self, op, block_idx, vis_num, mask_index=0, vis_num_pc=1, vis_vT=False,
                                                   pca_rank=50, edit_prompt=None, null_space_projection=False,
                                                   pca_rank_null=50):
        """edit.py:1045-1174: the text-supervised direction through the Jacobian of the decoded image, projected onto the
        null space of the complement-mask Jacobian; ``use_sega`` decodes with the three-branch guidance instead."""
        prep = self._prepare(mask_index)
        if prep is None:
            return None
        zt, t, t_idx, mask = prep
        F, E, N = self.for_prompt_emb, self.edit_prompt_emb, self.null_prompt_emb
        if self.use_sega:
            self.EXP_NAME = f'sega-edit_prompt-{self.edit_prompt}'
            return self.DDIMforwardsteps(zt, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                         null_prompt_emb=N, mode="null+(for-null)+(edit-null)")
        save_dir = os.path.join(self.result_folder, "basis",
                                f'local_basis-{self.edit_t}T-"{self.edit_prompt}"-pca-rank-{pca_rank}-select-mask{mask_index}')

        def modify():
            return None, self.get_delta_zt_via_grad(zt, t, t_idx, F, E, N, mask=mask, mode=self.tilda_v_score_type)
        vT = self._solve_or_load(save_dir, zt, t, t_idx, mask, op, block_idx, pca_rank, pca_rank_null, null_space_projection, modify)
        original_zt = zt.clone()
        for pc_idx in range(vis_num_pc):
            self.EXP_NAME = (f'Edit_zt-edit_{self.edit_t}T-{op}-block_{block_idx}-pc_{pc_idx:0=3d}_pos-edit_prompt-{self.edit_prompt}'
                             f'_select_mask{mask_index}_null_space_projection_{null_space_projection}_null_space_rank_'
                             f'{pca_rank_null}_{self.tilda_v_score_type}')
            zb = self._walk(original_zt, vT[pc_idx, :], vis_num)
        return self.DDIMforwardsteps(zb, t_start_idx=self.edit_t_idx, t_end_idx=-1, for_prompt_emb=F, edit_prompt_emb=E,
                                     null_prompt_emb=N, mode="null+(for-null)")

    run_edit_null_space_projection_xt = None
    run_edit_null_space_projection_xt_semantic = None
