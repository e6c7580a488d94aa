"""Generate tests/golden/tloco_tiny.pt by running the REFERENCE's ``EditDeepFloydIF`` methods (imported read-only from
/root/reference with the stub modules of make_golden.py) on a stand-in conditional denoiser, and pin
oracle/tloco_oracle.py against them.  Runs only in the build container; the fixture is data (inputs + expected outputs).

Stand-in denoiser (the IF U-Net itself is diffusers' UNet2DConditionModel, un-vendored): the reference's own
guided-diffusion ``UNetModel`` (P2 switches) with ``emb = time_embed(t) + cond_proj(mean_tokens(prompt_emb))`` -- the
forward loop of unet.py:656-676 with that one addition -- returning all 2C channels as ``.sample`` so that the
reference's learned-variance split (edit.py:1328-1336) is exercised.

    python oracle/make_golden_tloco.py
"""
from __future__ import annotations

import math
import os
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import make_golden as mg  # noqa: E402


class _Out:
    def __init__(self, sample):
        self.sample = sample


def ref_cond_unet(model, cond_w, cond_b):
    """unet(x, t, encoder_hidden_states=E).sample built from the reference's UNetModel modules."""
    import torch as th
    from models.guided_diffusion.nn import timestep_embedding

    def call(x, t, encoder_hidden_states=None, **kw):
        t = t.unsqueeze(0) if isinstance(t, th.Tensor) and len(t.shape) == 0 else t
        emb = model.time_embed(timestep_embedding(t, model.model_channels))
        emb = emb + th.nn.functional.linear(encoder_hidden_states.mean(dim=1), cond_w, cond_b)
        hs = []
        h = x
        for module in model.input_blocks:
            h = module(h, emb)
            hs.append(h)
        h = model.middle_block(h, emb)
        for module in model.output_blocks:
            h = th.cat([h, hs.pop()], dim=1)
            h = module(h, emb)
        return _Out(model.out(h))
    return call


def main(redit, cfg, out_name, mid=False):
    """`mid`: BASELINE config 5's size class -- 64x64 pixels, four levels, attention at 32 / 16 / 8 (1024-token level
    included) at a width the CPU reference finishes in minutes; CFG noise, get_x0 and the CFG-combined solver only."""
    from utils.utils import betas_for_alpha_bar, get_deepfloyd_if_scheduler
    import tloco_oracle as tl
    import loco_oracle as orc
    from loco_edit_amd.config import synth_params
    from loco_edit_amd.tloco import cond_params
    torch.set_num_threads(8)
    tmpdir = tempfile.mkdtemp(prefix="loco_golden_tloco_")
    D, NTOK = 16, 7
    params = synth_params(cfg, seed=0)
    cp = cond_params(cfg, D, seed=0)
    model = mg.ref_model_adm(cfg, params)
    cw, cb = torch.from_numpy(cp["cond_proj.weight"].copy()), torch.from_numpy(cp["cond_proj.bias"].copy())
    unet = ref_cond_unet(model, cw, cb)

    # ---- the reference object without its diffusers / T5 / SAM constructor work
    ed = object.__new__(redit.EditDeepFloydIF)
    sched = types.SimpleNamespace()
    betas = betas_for_alpha_bar(1000, lambda ts: math.cos((ts + 0.008) / 1.008 * math.pi / 2) ** 2)
    sched.betas = torch.tensor(betas, dtype=torch.float32)
    sched.alphas_cumprod = torch.cumprod(1.0 - sched.betas, dim=0)
    sched.scale_model_input = lambda x, t: x
    sargs = Namespace(use_yh_custom_scheduler=True, device=torch.device("cpu"), dtype=torch.float32)
    ed.scheduler = get_deepfloyd_if_scheduler(sargs, sched)
    ed.unet = unet
    ed.device, ed.dtype, ed.buffer_device, ed.memory_bound = torch.device("cpu"), torch.float32, "cpu", 50
    ed.for_steps, ed.use_yh_custom_scheduler = 100, True
    ed.guidance_scale, ed.guidance_scale_edit = 7.5, 4.0
    ed.result_folder, ed.EXP_NAME = tmpdir, "golden"
    ed.c_in, ed.image_size = cfg.in_channels, cfg.resolution
    ed.scheduler.set_timesteps(100, device="cpu")
    ed.edit_t = 0.6
    ed.edit_t_idx = (ed.scheduler.timesteps - 0.6 * 1000).abs().argmin()
    g = torch.Generator().manual_seed(31)
    for_e, edit_e, null_e = (torch.randn(1, NTOK, D, generator=g) for _ in range(3))
    ed.for_prompt_emb, ed.edit_prompt_emb, ed.null_prompt_emb = for_e, edit_e, null_e
    ed.tilda_v_score_type = "null+(for-null)+(edit-null)"

    po = orc.to_torch(params)
    po.update({k: torch.from_numpy(v.copy()) for k, v in cp.items()})
    ot = tl.OracleTLoco(po, cfg, guidance_scale=7.5, guidance_scale_edit=4.0)
    assert torch.equal(ot.sched.alphas_cumprod, sched.alphas_cumprod), "schedule restatement differs"
    ot.sched.set_timesteps(100)
    assert torch.equal(ot.sched.timesteps, ed.scheduler.timesteps)
    assert int(ed.edit_t_idx) == ot.edit_t_idx

    out = {"cfg": dict(resolution=cfg.resolution, ch=cfg.ch, ch_mult=tuple(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                       attn_resolutions=tuple(cfg.attn_resolutions), arch=cfg.arch, num_head_channels=cfg.num_head_channels,
                       learn_sigma=cfg.learn_sigma, gn_eps=cfg.gn_eps),
           "weights_seed": 0, "cond_dim": D, "for_e": for_e, "edit_e": edit_e, "null_e": null_e,
           "guidance_scale": 7.5, "guidance_scale_edit": 4.0, "alphas_cumprod": sched.alphas_cumprod.clone(),
           "timesteps": ed.scheduler.timesteps.clone(), "edit_t_idx": int(ed.edit_t_idx)}
    gx = torch.Generator().manual_seed(1)
    x = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=gx)
    t = ed.scheduler.timesteps[int(ed.edit_t_idx)]
    out["x"], out["t"] = x, t.clone()
    r = cfg.resolution // 32
    mask = mg.rect_mask(cfg, 12 * r, 20 * r, 8 * r, 18 * r)
    out["mask"] = mask

    # ---- 1. CFG noise, every mode (edit.py:1286-1373) on a batch of 2
    xb = torch.cat([x, x.flip(-1)], dim=0)
    out["eps_modes"] = {}
    with torch.no_grad():
        for mode in (tl.MODES if not mid else ("null+(for-null)+(edit-null)",)):
            e_ref = ed._classifer_free_guidance(xb, t, for_e, edit_e, null_e, mode, True)
            mg.check(f"tloco/cfg[{mode}]", ot.cfg_noise(xb, t, for_e, edit_e, null_e, mode), e_ref)
            out["eps_modes"][mode] = e_ref
        e_nocfg = ed._classifer_free_guidance(xb, t, for_e, edit_e, null_e, "null+(for-null)", False)
        mg.check("tloco/no-cfg", ot.cfg_noise(xb, t, for_e, edit_e, null_e, "null+(for-null)", do_cfg=False), e_nocfg)
        out["eps_nocfg"] = e_nocfg
        # ---- 2. get_x0 (edit.py:1566-1587)
        x0m = ed.get_x0(x, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=mask, mode="null+(for-null)+(edit-null)")
        mg.check("tloco/get_x0", ot.get_x0(x, t, for_e, edit_e, null_e, mask=mask), x0m)
        out["x0_masked"] = x0m

    # ---- 3. solver (edit.py:1589-1676), V0 injected
    gv = torch.Generator().manual_seed(7)
    v0 = torch.randn(cfg.n, 3, generator=gv)
    out["v0"] = v0
    real_randn = torch.randn

    def fake_randn(*size, **kw):
        if len(size) == 2 and size[0] == cfg.n:
            return v0[:, :size[1]].clone()
        return real_randn(*size, **kw)
    out["solver"] = {}
    specs = ((("null+(for-null)", 6, mask), ("null+(for-null)+(edit-null)", 3, ~mask), ("(for-edit)", 3, None)) if not mid
             else (("null+(for-null)+(edit-null)", 4, mask), ("null+(for-null)", 3, ~mask)))
    for mode, n_iter, msk in specs:
        torch.randn = fake_randn
        try:
            with torch.no_grad():
                u, s, vT = ed.local_encoder_decoder_pullback_xt(x, t, ed.edit_t_idx, for_e, edit_e, null_e, pca_rank=3,
                                                                chunk_size=5, min_iter=n_iter, max_iter=n_iter,
                                                                convergence_threshold=1e-3, mask=msk, mode=mode)
        finally:
            torch.randn = real_randn
        ou, os_, ovT = ot.pullback(x, t, for_e, edit_e, null_e, 3, v0, min_iter=n_iter, max_iter=n_iter, mask=msk, mode=mode)
        mg.check(f"tloco/solver[{mode}] s", os_, s, rtol=1e-3)
        c = mg.abs_cos_rows(ovT, vT)
        print(f"  oracle vs reference [tloco/solver {mode}] |cos| min {c.min().item():.6f}")
        assert c.min() > 0.999
        out["solver"][mode] = {"n_iter": n_iter, "mask": msk, "u": u, "s": s, "vT": vT}

    if mid:
        torch.save(out, os.path.join(GOLD, out_name))
        print("done ->", os.path.join(GOLD, out_name))
        return
    # ---- 4. direction through the Jacobian (edit.py:1680-1717) and 5. direct directions (:1720-1741)
    v_grad = ed.get_delta_xt_via_grad(x, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=mask,
                                      mode="null+(for-null)+(edit-null)").detach()
    mg.check("tloco/delta_xt_via_grad", ot.delta_xt_via_grad(x, t, for_e, edit_e, null_e, mask=mask), v_grad, rtol=1e-3)
    out["v_grad"] = v_grad
    out["v_direct"] = {}
    with torch.no_grad():
        for mode in ("(for-edit)-direct", "(edit-null)-direct", "proj_null[for-null](edit-null)-direct"):
            vd = ed.get_v_modify(x, t, ed.edit_t_idx, for_e, edit_e, null_e, mask=mask, mode=mode, jacobian=False)
            mg.check(f"tloco/v_modify[{mode}]", ot.v_modify_direct(x, t, for_e, edit_e, null_e, mode), vd)
            out["v_direct"][mode] = vd
        # ---- 6. sampler: x_T -> x_t at the edit step, then x_t -> x_0 of a 2-image batch (edit.py:1412-1481)
        gT = torch.Generator().manual_seed(5)
        xT = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=gT)
        xt_e, t_e, i_e = ed.DDPMforwardsteps(xT, t_start_idx=0, t_end_idx=ed.edit_t_idx, for_prompt_emb=for_e,
                                             edit_prompt_emb=edit_e, null_prompt_emb=null_e, mode="null+(for-null)")
        oxt, ott, oi = ot.forwardsteps(xT, 0, ot.edit_t_idx, for_e, edit_e, null_e)
        assert int(i_e) == oi
        mg.check("tloco/to_t", oxt, xt_e, rtol=1e-3, atol=1e-4)
        xb2 = torch.cat([xt_e, xt_e + 0.25 * x], dim=0)
        ed.memory_bound = 2      # batch 2 with CFG -> chunk(2 // (2 // 2)) (the reference's chunking divides by zero for
        #                          1 < batch < memory_bound // 2, edit.py:1455)
        img = ed.DDPMforwardsteps(xb2, t_start_idx=ed.edit_t_idx, t_end_idx=-1, for_prompt_emb=for_e,
                                  edit_prompt_emb=edit_e, null_prompt_emb=null_e, mode="null+(for-null)+(edit-null)")
        odec = ot.forwardsteps(xb2, ot.edit_t_idx, -1, for_e, edit_e, null_e, mode="null+(for-null)+(edit-null)")
        oimg = ((odec / 2 + 0.5).clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1)
        diff = (oimg.int() - img.int()).abs()
        print(f"  oracle vs reference [tloco/decode uint8] max |diff| {int(diff.max())}, differing pixels {int((diff > 0).sum())}")
        assert int(diff.max()) <= 1
        out.update(xT=xT, xt_edit=xt_e, t_edit=t_e.clone(), dec_in=xb2, dec_u8=img, dec_f32=odec)
    torch.save(out, os.path.join(GOLD, out_name))
    print("done ->", os.path.join(GOLD, out_name))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["tiny", "mid"], default=None)
    a = ap.parse_args()
    redit_ = mg.import_reference()[0]
    from loco_edit_amd.config import TINY_ADM, MID_IF64
    if a.only in (None, "tiny"):
        main(redit_, TINY_ADM, "tloco_tiny.pt")
    if a.only in (None, "mid"):
        main(redit_, MID_IF64, "tloco_mid.pt", mid=True)
