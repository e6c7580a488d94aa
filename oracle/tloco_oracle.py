"""TEST INFRASTRUCTURE -- CPU restatement of the reference's text-supervised T-LOCO orchestration in pixel space
(DeepFloyd-IF stage-I path, reference ``src/modules/edit.py`` class ``EditDeepFloydIF``), used only by tests/,
``__graft_entry__`` and the golden generator.  The product path never imports it.

What is restated (each function cites the lines it follows): the classifier-free-guidance combination of the
conditional noise predictions with the learned-variance channels split off (``_classifer_free_guidance``
:1286-1373), ``get_x0`` (:1566-1587), the CFG-combined PMP-Jacobian subspace iteration
(``local_encoder_decoder_pullback_xt`` :1589-1676), the edit direction through the Jacobian
(``get_delta_xt_via_grad`` :1680-1717) and directly in noise space (``get_v_modify`` :1720-1741), the DDIM sampler
with CFG (``DDPMforwardsteps`` :1412-1481; scheduler ``step`` utils.py:187-213, t_max 990) and the squared-cosine
schedule DeepFloyd-IF ships (utils.py:425-441 ``betas_for_alpha_bar``).

The denoiser is a stand-in: the conditional U-Net of the reference is diffusers' ``UNet2DConditionModel`` with the IF
weights (un-vendored, ``requirements.txt:4``), so parity of the *architecture* stays unpinned.  Here the duck type
``unet(x, t, encoder_hidden_states=E).sample -> [B, 2C, H, W]`` is filled by the guided-diffusion U-Net of
``loco_oracle`` whose time embedding receives ``cond_proj(mean_tokens(E))`` -- the slot diffusers uses for
``addition_embed_type="text"`` and guided-diffusion for its class embedding.  The golden generator runs the
reference's own ``EditDeepFloydIF`` methods on the same stand-in built from the reference's modules.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

import loco_oracle as orc

MODES = ("null+(for-null)+(edit-null)", "null+(for-null)", "null+(edit-null)", "(for-edit)", "(for-null)", "(edit-null)")


def squaredcos_alphas_cumprod(n: int = 1000, max_beta: float = 0.999) -> torch.Tensor:
    """utils.py:425-441 betas_for_alpha_bar with the cosine alpha_bar; float32 like the diffusers scheduler."""
    def alpha_bar(ts):
        return math.cos((ts + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = [min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)]
    betas = torch.tensor(betas, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


class IFScheduler:
    """The monkey-patched IF scheduler of utils.py:159-213: float timesteps linspace(0,1,N)*990, DDIM update."""
    t_max = 990

    def __init__(self):
        self.alphas_cumprod = squaredcos_alphas_cumprod()
        self.timesteps = self.timesteps_next = None

    def set_timesteps(self, n: int):
        seq = torch.linspace(0, 1, n) * self.t_max
        seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]], dim=0)
        self.timesteps = torch.flip(seq[1:], dims=[0])
        self.timesteps_next = torch.flip(seq_prev[1:], dims=[0])

    def alpha_at(self, t) -> torch.Tensor:
        return self.alphas_cumprod[int(torch.as_tensor(t).long())]       # extract(): gather at t.long() (utils.py:458)

    def step(self, et, t, xt):
        idx = self.timesteps.tolist().index(float(t))
        at, at_next = self.alpha_at(t), self.alpha_at(self.timesteps_next[idx])
        p_xt = (xt - et * (1 - at).sqrt()) / at.sqrt()
        return at_next.sqrt() * p_xt + (1 - at_next).sqrt() * et


def cond_embedding(p: Dict[str, torch.Tensor], prompt_emb: torch.Tensor) -> torch.Tensor:
    """[B, tokens, D] -> [B, 4*ch]: mean over tokens, then the ``cond_proj`` Linear of the stand-in."""
    return F.linear(prompt_emb.mean(dim=1), p["cond_proj.weight"], p["cond_proj.bias"])


class OracleTLoco:
    def __init__(self, params, cfg, guidance_scale=7.5, guidance_scale_edit=4.0, for_steps=100, edit_t=0.6):
        self.p, self.cfg = params, cfg
        self.guidance_scale, self.guidance_scale_edit = guidance_scale, guidance_scale_edit
        self.sched = IFScheduler()
        self.for_steps = for_steps
        self.sched.set_timesteps(for_steps)
        self.edit_t_idx = int((self.sched.timesteps - edit_t * 1000).abs().argmin())

    def unet_full(self, x, t, prompt_emb):
        if getattr(self.cfg, "context_dim", 0) > 0:       # text through the cross-attention stages
            return orc.unet_forward_adm(self.p, self.cfg, x, t, full=True, context=prompt_emb)
        return orc.unet_forward_adm(self.p, self.cfg, x, t, emb_add=cond_embedding(self.p, prompt_emb), full=True)

    # -- edit.py:1286-1373
    def cfg_noise(self, x, t, for_e, edit_e, null_e, mode, do_cfg=True):
        c = x.shape[1]
        B = x.shape[0]
        def eps(e):
            return self.unet_full(x, t, e.repeat(B, 1, 1))[:, :c]      # learned variance split off (:1328-1336)
        if not do_cfg:
            return self.unet_full(x, t, for_e.repeat(B, 1, 1))        # :1317-1318 (all 2C channels, as the reference)
        g, ge = self.guidance_scale, self.guidance_scale_edit
        if mode == "null+(for-null)+(edit-null)":
            n = eps(null_e)
            return n + g * (eps(for_e) - n) + ge * (eps(edit_e) - n)
        if mode == "null+(for-null)":
            n = eps(null_e)
            return n + g * (eps(for_e) - n)
        if mode == "null+(edit-null)":
            n = eps(null_e)
            return n + g * (eps(edit_e) - n)
        if mode == "(for-edit)":
            return g * (eps(for_e) - eps(edit_e))
        if mode == "(for-null)":
            return g * (eps(for_e) - eps(null_e))
        if mode == "(edit-null)":
            return g * (eps(edit_e) - eps(null_e))
        raise ValueError(mode)

    # -- edit.py:1566-1587
    def get_x0(self, xt, t, for_e, edit_e, null_e, mask=None, mode="null+(for-null)+(edit-null)", flatten=False):
        eps = self.cfg_noise(xt, t, for_e, edit_e, null_e, mode, do_cfg=self.guidance_scale > 1.0)
        at = self.sched.alpha_at(t)
        x0 = (xt - eps * (1 - at).sqrt()) / at.sqrt()
        if mask is not None:
            return x0[:, mask]
        return x0.reshape(x0.shape[0], -1) if flatten else x0

    # -- edit.py:1589-1676 (V0 injected)
    def pullback(self, xt, t, for_e, edit_e, null_e, pca_rank, v0, min_iter=10, max_iter=100,
                 convergence_threshold=1e-3, mask=None, mode="null+(for-null)+(edit-null)", chunk_size=25):
        c, hh, ww = xt.shape[1:]
        n = c * hh * ww
        num_chunk = pca_rank // chunk_size if pca_rank % chunk_size == 0 else pca_rank // chunk_size + 1
        a = torch.tensor(0.0)
        v = torch.linalg.qr(v0.float())[0].T.reshape(-1, c, hh, ww)
        for i in range(max_iter):
            v_prev = v.detach().clone()
            u = []
            for vi in v.chunk(num_chunk):
                g = lambda al: self.get_x0(xt + al * vi, t, for_e, edit_e, null_e, mask=mask, mode=mode)
                u.append(torch.func.jacfwd(g, argnums=0, randomness="error")(a).detach())
            u = torch.cat(u, dim=0)
            if mask is None:
                g2 = lambda x_: torch.einsum("bcwh,icwh->b", u, self.get_x0(x_, t, for_e, edit_e, null_e, mask=mask, mode=mode))
            else:
                g2 = lambda x_: torch.einsum("bl,il->b", u, self.get_x0(x_, t, for_e, edit_e, null_e, mask=mask, mode=mode))
            v_ = torch.autograd.functional.jacobian(g2, xt).reshape(-1, n).float()
            _, s, v = torch.linalg.svd(v_, full_matrices=False)
            v = v.reshape(-1, c, hh, ww)
            if torch.allclose(v_prev, v, atol=convergence_threshold) and i > min_iter:
                break
        L = n if mask is None else int(mask.sum())
        return u.reshape(-1, L).T.detach(), s.sqrt().detach(), v.reshape(-1, n).detach()

    # -- edit.py:1680-1717: direction = normalised J^T (x0_hat[mode] - x0_hat["null+(for-null)"])
    def delta_xt_via_grad(self, xt, t, for_e, edit_e, null_e, mask=None, mode="null+(for-null)+(edit-null)"):
        do = self.guidance_scale > 1.0
        at = self.sched.alpha_at(t)
        e0 = self.cfg_noise(xt, t, for_e, edit_e, null_e, "null+(for-null)", do)
        e1 = self.cfg_noise(xt, t, for_e, edit_e, null_e, mode, do)
        d = (xt - e1 * (1 - at).sqrt()) / at.sqrt() - (xt - e0 * (1 - at).sqrt()) / at.sqrt()
        dflat = d[:, mask] if mask is not None else d.reshape(d.shape[0], -1)
        g = lambda v: torch.sum(dflat * self.get_x0(v, t, for_e, edit_e, null_e, mask=mask, mode=mode, flatten=True))
        v_ = torch.autograd.functional.jacobian(g, xt).reshape(-1, xt[0].numel())
        return v_ / v_.norm(dim=1, keepdim=True)

    # -- edit.py:1720-1741 (the three direct modes)
    def v_modify_direct(self, xt, t, for_e, edit_e, null_e, mode):
        if mode == "(for-edit)-direct":
            return self.cfg_noise(xt, t, for_e, edit_e, null_e, "(for-edit)").reshape(1, -1)
        if mode == "(edit-null)-direct":
            return -self.cfg_noise(xt, t, for_e, edit_e, null_e, "(edit-null)").reshape(1, -1)
        if mode == "proj_null[for-null](edit-null)-direct":
            e1 = self.cfg_noise(xt, t, for_e, edit_e, null_e, "(for-null)").reshape(1, -1)
            e2 = self.cfg_noise(xt, t, for_e, edit_e, null_e, "(edit-null)").reshape(1, -1)
            return -(e2 - ((e2 * e1).sum() / (e1 * e1).sum()) * e1)
        raise ValueError(mode)

    # -- edit.py:1412-1481 (eta = 0; returns x_t at t_end_idx or the final sample before the uint8 conversion)
    @torch.no_grad()
    def forwardsteps(self, xt, t_start_idx, t_end_idx, for_e, edit_e, null_e, mode="null+(for-null)"):
        self.sched.set_timesteps(self.for_steps)
        for t_idx, t in enumerate(self.sched.timesteps):
            if t_idx < t_start_idx:
                continue
            elif t_start_idx == t_idx:
                pass
            elif t_idx == t_end_idx:
                return xt, t, t_idx
            eps = self.cfg_noise(xt, t, for_e, edit_e, null_e, mode, do_cfg=self.guidance_scale > 1.0)
            xt = self.sched.step(eps, t, xt)
        return xt
