"""TEST INFRASTRUCTURE -- CPU restatement of the reference's text-supervised T-LOCO orchestration in LATENT space
(Stable Diffusion path, reference ``src/modules/edit.py`` class ``EditStableDiffusion`` :483-1196), used only by tests/
and the golden generator.  The product path never imports it.

Restated (each function cites the lines it follows): the latent DDIM inversion with its ``vae.encode`` (:568-633),
``get_x0`` with the decode (:757-781), the subspace iteration on
the decoded image's Jacobian (``local_encoder_decoder_pullback_zt`` :830-915), the direction through the Jacobian
(``get_delta_zt_via_grad`` :784-828), the sampler with the final decode (``DDIMforwardsteps`` :677-754) and the
``scaled_linear`` alpha-bar table of the pipeline scheduler the reference patches (utils.py:147-157).  The CFG
combination is the one of ``tloco_oracle`` (edit.py:636-674 is the pixel-space function without the variance split).

Both networks are stand-ins (diffusers' UNet2DConditionModel and AutoencoderKL are un-vendored): the guided-diffusion
U-Net of ``loco_oracle`` on 4 latent channels with ``cond_proj`` conditioning, and ``loco_oracle.decoder_forward``.
The golden generator runs the reference's own ``EditStableDiffusion`` methods on stand-ins built from the reference's
modules (guided_diffusion ``UNetModel``; DDPM ``ResnetBlock`` / ``AttnBlock`` / ``Upsample`` for the decoder).
"""
from __future__ import annotations

import torch

import loco_oracle as orc
import tloco_oracle as tl

LATENT_SCALE = 0.18215


def scaled_linear_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class SDScheduler(tl.IFScheduler):
    t_max = 999

    def __init__(self):
        self.alphas_cumprod = scaled_linear_alphas_cumprod()
        self.timesteps = self.timesteps_next = None

    def set_inversion_timesteps(self, n: int):
        """utils.py:172-179 (`is_inversion=True`): ascending float timesteps shifted by 1e-6, the next one as target."""
        seq = torch.linspace(0, 1, n) * self.t_max + 1e-6
        self.timesteps, self.timesteps_next = seq[:-1].clone(), seq[1:].clone()


def posterior_sample(moments: torch.Tensor, noise) -> torch.Tensor:
    """diffusers DiagonalGaussianDistribution.sample() (un-vendored; vae.py of the pinned diffusers): mean | logvar
    chunks, logvar clamped to [-30, 20], mean + exp(0.5 logvar) * noise.  noise None: the mean."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mean if noise is None else mean + std * noise


class OracleTLocoSD(tl.OracleTLoco):
    def __init__(self, params, cfg, dec_params, dec_cfg, guidance_scale=7.5, guidance_scale_edit=4.0, for_steps=100, edit_t=0.7):
        super().__init__(params, cfg, guidance_scale, guidance_scale_edit, for_steps, edit_t)
        self.dp, self.dcfg = dec_params, dec_cfg
        self.sched = SDScheduler()
        self.sched.set_timesteps(for_steps)
        self.edit_t_idx = int((self.sched.timesteps - edit_t * 1000).abs().argmin())

    def decode(self, z_scaled):
        return orc.decoder_forward(self.dp, self.dcfg, z_scaled)

    # -- edit.py:568-633 (run_DDIMinversion): z0 = vae.encode(x0).latent_dist.sample() * 0.18215, then the ascending DDIM
    #    loop on the inversion prompt (classifier-free guidance against the null prompt only when `guidance` is given)
    def inversion(self, x0, noise, enc_params, enc_cfg, inv_e, null_e, inv_steps, guidance=None, return_z0=False):
        z0 = posterior_sample(orc.encoder_forward(enc_params, enc_cfg, x0), noise) * LATENT_SCALE
        do_cfg = (self.guidance_scale > 1.0) and (guidance is not None)
        sched = SDScheduler()
        sched.set_inversion_timesteps(inv_steps)
        z = z0
        B = z.shape[0]
        for i, t in enumerate(sched.timesteps):
            if i == len(sched.timesteps) - 1:
                break
            if do_cfg:
                n = self.unet_full(z, t, null_e.repeat(B, 1, 1))
                eps = n + self.guidance_scale * (self.unet_full(z, t, inv_e.repeat(B, 1, 1)) - n)
            else:
                eps = self.unet_full(z, t, inv_e.repeat(B, 1, 1))
            z = sched.step(eps, t, z)
        return (z, z0) if return_z0 else z

    # -- edit.py:757-781
    def get_x0(self, zt, t, for_e, edit_e, null_e, mask=None, mode="null+(for-null)+(edit-null)", flatten=False):
        eps = self.cfg_noise(zt, t, for_e, edit_e, null_e, mode, do_cfg=self.guidance_scale > 1.0)
        at = self.sched.alpha_at(t)
        z0 = (zt - eps * (1 - at).sqrt()) / at.sqrt()
        x0 = self.decode(1 / LATENT_SCALE * z0)
        if mask is not None:
            return x0[:, mask]
        return x0.reshape(x0.shape[0], -1) if flatten else x0

    # -- edit.py:830-915 (V0 injected)
    def pullback(self, zt, t, for_e, edit_e, null_e, pca_rank, v0, min_iter=10, max_iter=100,
                 convergence_threshold=1e-3, mask=None, mode="null+(for-null)+(edit-null)", chunk_size=25):
        c, hh, ww = zt.shape[1:]
        n = c * hh * ww
        num_chunk = pca_rank // chunk_size if pca_rank % chunk_size == 0 else pca_rank // chunk_size + 1
        a = torch.tensor(0.0)
        v = torch.linalg.qr(v0.float())[0].T.reshape(-1, c, hh, ww)
        for i in range(max_iter):
            v_prev = v.detach().clone()
            u = []
            for vi in v.chunk(num_chunk):
                g = lambda al: self.get_x0(zt + al * vi, t, for_e, edit_e, null_e, mask=mask, mode=mode, flatten=mask is None)
                u.append(torch.func.jacfwd(g, argnums=0, randomness="error")(a).detach())
            u = torch.cat(u, dim=0)
            g2 = lambda z_: torch.einsum("bl,il->b", u, self.get_x0(z_, t, for_e, edit_e, null_e, mask=mask, mode=mode,
                                                                        flatten=mask is None))
            v_ = torch.autograd.functional.jacobian(g2, zt).reshape(-1, n).float()
            _, s, v = torch.linalg.svd(v_, full_matrices=False)
            v = v.reshape(-1, c, hh, ww)
            if torch.allclose(v_prev, v, atol=convergence_threshold) and i > min_iter:
                break
        return u.reshape(u.shape[0], -1).T.detach(), s.sqrt().detach(), v.reshape(-1, n).detach()

    # -- edit.py:784-828: normalised J_mode^T (decoded x0_hat[mode] - decoded x0_hat["null+(for-null)"]) on the mask
    def delta_zt_via_grad(self, zt, t, for_e, edit_e, null_e, mask, mode="null+(for-null)+(edit-null)"):
        with torch.no_grad():
            d = self.get_x0(zt, t, for_e, edit_e, null_e, mode=mode) - self.get_x0(zt, t, for_e, edit_e, null_e, mode="null+(for-null)")
        dflat = d[:, mask]
        g = lambda v: torch.sum(dflat * self.get_x0(v, t, for_e, edit_e, null_e, mask=mask, mode=mode, flatten=True))
        v_ = torch.autograd.functional.jacobian(g, zt).reshape(-1, zt[0].numel())
        return v_ / v_.norm(dim=1, keepdim=True)

    # -- edit.py:747-754: the tail of DDIMforwardsteps
    @torch.no_grad()
    def decode_final(self, latents):
        latents = 1 / LATENT_SCALE * latents
        x0 = (self.decode(latents) / 2 + 0.5).clamp(0, 1)
        return latents, (x0 * 255).to(torch.uint8).permute(0, 2, 3, 1), x0
