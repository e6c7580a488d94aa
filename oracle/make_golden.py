"""Generate tests/golden/*.pt by running the REFERENCE itself (imported read-only
from /root/reference with stub modules for the packages this image lacks) and
pin oracle/loco_oracle.py against it.

Runs only in the build container (the GPU box has no /root/reference); the
fixtures it writes are data (inputs + expected outputs), no reference source.

    python oracle/make_golden.py [--full]     # --full adds the 256x256 summaries
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    """Stub the absent third-party modules and import."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", save_image=lambda *a, **k: None)
    tv.transforms = stub("torchvision.transforms", Compose=_Dummy, ToTensor=_Dummy, Normalize=_Dummy,
                         Resize=_Dummy, CenterCrop=_Dummy, ToPILImage=_Dummy, InterpolationMode=_Dummy)
    df = stub("diffusers", DDIMScheduler=_Dummy, DDIMPipeline=_Dummy, StableDiffusionPipeline=_Dummy,
              DiffusionPipeline=_Dummy, LCMScheduler=_Dummy, AutoPipelineForText2Image=_Dummy)
    df.utils = stub("diffusers.utils", pt_to_pil=lambda *a, **k: None)
    stub("skimage")
    stub("transformers", pipeline=lambda *a, **k: None)
    sys.path.insert(0, REF)
    import modules.edit as redit  # noqa
    from utils.utils import YHCustomScheduler, extract  # noqa
    from models.ddpm.diffusion import PullBackDDPM  # noqa
    return redit, YHCustomScheduler, extract, PullBackDDPM


def ref_model(PullBackDDPM, cfg, params):
    conf = Namespace(
        model=Namespace(ch=cfg.ch, out_ch=cfg.out_ch, ch_mult=list(cfg.ch_mult),
                        num_res_blocks=cfg.num_res_blocks, attn_resolutions=list(cfg.attn_resolutions),
                        dropout=0.0, in_channels=cfg.in_channels, resamp_with_conv=True),
        data=Namespace(image_size=cfg.resolution))
    args = Namespace(config=conf, device=torch.device("cpu"), dtype=torch.float32)
    m = PullBackDDPM(args)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    m.requires_grad_(False)
    return m


def ref_model_adm(cfg, params):
    """guided_diffusion UNetModel with the P2 switches (script_util.py:166-190, 379-435)."""
    from models.guided_diffusion.unet import UNetModel
    ds = tuple(cfg.resolution // r for r in cfg.attn_resolutions)
    m = UNetModel(image_size=cfg.resolution, in_channels=3, model_channels=cfg.ch,
                  out_channels=6 if cfg.learn_sigma else 3, num_res_blocks=cfg.num_res_blocks,
                  attention_resolutions=ds, dropout=0, channel_mult=tuple(cfg.ch_mult),
                  num_head_channels=cfg.num_head_channels, use_scale_shift_norm=bool(getattr(cfg, "scale_shift_norm", True)),
                  resblock_updown=bool(getattr(cfg, "resblock_updown", True)))
    m.device = torch.device("cpu")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    m.eval()
    m.requires_grad_(False)
    return m


def ref_edit(redit, YHCustomScheduler, model, tmpdir):
    """Build EditUncondDiffusion without its HF/SAM/dataset constructor work."""
    ed = object.__new__(redit.EditUncondDiffusion)
    sargs = Namespace(noise_schedule="linear", device=torch.device("cpu"), dtype=torch.float32)
    ed.unet = model
    ed.scheduler = YHCustomScheduler(sargs)
    ed.device = torch.device("cpu")
    ed.dtype = torch.float32
    ed.for_steps = 100
    ed.inv_steps = 100
    ed.use_yh_custom_scheduler = True
    ed.buffer_device = "cpu"
    ed.memory_bound = 50
    ed.result_folder = tmpdir
    ed.EXP_NAME = "golden"
    ed.dataset_name = "CelebA_HQ_mask"
    ed.scheduler.set_timesteps(100, device="cpu")
    ed.edit_t_idx = (ed.scheduler.timesteps - 0.6 * 1000).abs().argmin()
    ed.performance_boosting_t_idx = (ed.scheduler.timesteps - 0.2 * 1000).abs().argmin()
    return ed


def rect_mask(cfg, r0, r1, c0, c1):
    m = torch.zeros(cfg.in_channels, cfg.resolution, cfg.resolution, dtype=torch.bool)
    m[:, r0:r1, c0:c1] = True
    return m


def check(name, a, b, rtol=1e-4, atol=1e-5):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol * max(1.0, ref))
    print(f"  oracle vs reference [{name}]: max abs err {err:.3e} (ref max {ref:.3e}) {'OK' if ok else 'MISMATCH'}")
    if not ok:
        raise SystemExit(f"oracle restatement disagrees with the reference on {name}")


def abs_cos_rows(a, b):
    a, b = a.double(), b.double()
    a = a / a.norm(dim=1, keepdim=True)
    b = b / b.norm(dim=1, keepdim=True)
    return (a * b).sum(dim=1).abs()


def gen_for_config(tag, cfg, redit, YHS, PullBackDDPM, k, k_null, n_iter, mrect, tmpdir,
                   full_tensors=True, pipeline=True, oracle_solver=True):
    import loco_oracle as orc
    from loco_edit_amd.config import synth_params
    torch.manual_seed(0)
    params = synth_params(cfg, seed=0)
    model = ref_model_adm(cfg, params) if cfg.arch == "adm" else ref_model(PullBackDDPM, cfg, params)
    ed = ref_edit(redit, YHS, model, tmpdir)
    oed = orc.OracleEdit(orc.to_torch(params), cfg)
    out = {"cfg": dict(resolution=cfg.resolution, ch=cfg.ch, ch_mult=tuple(cfg.ch_mult),
                       num_res_blocks=cfg.num_res_blocks, attn_resolutions=tuple(cfg.attn_resolutions),
                       arch=cfg.arch, num_head_channels=cfg.num_head_channels, learn_sigma=cfg.learn_sigma,
                       gn_eps=cfg.gn_eps, scale_shift_norm=getattr(cfg, "scale_shift_norm", True),
                       resblock_updown=getattr(cfg, "resblock_updown", True)),
           "weights_seed": 0}
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=g)
    ed.scheduler.set_timesteps(100, device="cpu")
    t = ed.scheduler.timesteps[int(ed.edit_t_idx)]
    out["x"], out["t"] = x, t.clone()

    # --- fixture family 2: denoiser forward
    with torch.no_grad():
        eps_ref = model(x, t)
        eps_orc = oed.unet(x, t)
    check(f"{tag}/unet_forward", eps_orc, eps_ref)
    out["eps"] = eps_ref if full_tensors else None
    gs = torch.Generator().manual_seed(11)
    sidx = torch.randint(0, eps_ref.numel(), (4096,), generator=gs)
    out["eps_sample_idx"] = sidx
    out["eps_sample"] = eps_ref.reshape(-1)[sidx].clone()
    out["eps_sum"] = eps_ref.double().sum().item()
    out["eps_sqsum"] = (eps_ref.double() ** 2).sum().item()

    # --- fixture family 3: J V and U^T J
    mask = rect_mask(cfg, *mrect)
    out["mask"] = mask
    gv = torch.Generator().manual_seed(7)
    v0 = torch.randn(cfg.n, k, generator=gv)
    out["v0"] = v0 if full_tensors else None
    out["v0_seed"] = 7
    q, _ = torch.linalg.qr(v0)
    V = q.T.reshape(-1, *x.shape[1:]).contiguous()
    a = torch.tensor(0.0)
    gfun = lambda a_: ed.get_x0(t, x + a_ * V, mask=mask)
    U_ref = torch.func.jacfwd(gfun)(a).detach()
    U_orc = orc.jvp_x0(oed, x, t, V, mask=mask)
    check(f"{tag}/JV", U_orc, U_ref)
    from einops import einsum
    g2 = lambda x_: einsum(U_ref, ed.get_x0(t, x_, mask=mask), "b l, i l -> b")
    A_ref = torch.autograd.functional.jacobian(g2, x).reshape(k, -1).detach()
    A_orc = orc.vjp_x0(oed, x, t, U_ref, mask=mask)
    check(f"{tag}/UtJ", A_orc, A_ref)
    if full_tensors:
        out["V"], out["JV"], out["UtJ"] = V, U_ref, A_ref
    else:
        out["JV"] = U_ref  # [k, L] is small
        gp = torch.Generator().manual_seed(13)
        P = torch.randn(cfg.n, 64, generator=gp)
        out["UtJ_proj_seed"] = 13
        out["UtJ_proj"] = A_ref @ P
        out["UtJ_norm"] = A_ref.norm(dim=1)

    # --- fixture family 4: solver with injected V0, fixed iteration count
    if n_iter > 0:
        real_randn = torch.randn

        def fake_randn(*size, **kw):
            if len(size) == 2 and size[0] == cfg.n:
                return v0[:, :size[1]].clone()
            return real_randn(*size, **kw)

        torch.randn = fake_randn
        try:
          # the reference calls the solver under @torch.no_grad() (edit.py:2215)
          with torch.no_grad():
            u_m, s_m, vT_m = ed.local_encoder_decoder_pullback_xt(
                x=x, t=t, pca_rank=k, min_iter=n_iter, max_iter=n_iter,
                convergence_threshold=1e-4, mask=mask)
            if k_null:
                u_n, s_n, vT_n = ed.local_encoder_decoder_pullback_xt(
                    x=x, t=t, pca_rank=k_null, min_iter=n_iter, max_iter=n_iter,
                    convergence_threshold=1e-4, mask=~mask)
        finally:
            torch.randn = real_randn
        if oracle_solver:
            ou, os_, ovT, _ = oed.pullback(x, t, k, v0[:, :k], min_iter=n_iter, max_iter=n_iter,
                                           convergence_threshold=1e-4, mask=mask)
            check(f"{tag}/solver s", os_, s_m, rtol=1e-3)
            c = abs_cos_rows(ovT, vT_m)
            print(f"  oracle vs reference [{tag}/solver vT] |cos| min {c.min().item():.6f}")
            assert c.min() > 0.999
        out["n_iter"] = n_iter
        out["s_modify"] = s_m
        out["u_modify"] = u_m
        if full_tensors:
            out["vT_modify"] = vT_m
        else:
            gp = torch.Generator().manual_seed(17)
            P = torch.randn(cfg.n, 64, generator=gp)
            out["vT_proj_seed"] = 17
            out["vT_modify_proj"] = vT_m @ P
            out["vT_modify_f16"] = vT_m.to(torch.float16)
        if k_null:
            out["s_null"] = s_n
            if full_tensors:
                out["vT_null"] = vT_n
            # --- fixture family 5: projection + edit batch (edit.py:2317-2323, 2339-2363)
            vT_null = vT_n[:k_null, :]
            vT = (vT_null.T @ (vT_null @ vT_m.T)).T
            vT = vT_m - vT
            vT = vT / vT.norm(dim=1, keepdim=True)
            check(f"{tag}/project", orc.OracleEdit.project(vT_m, vT_n, k_null), vT)
            if full_tensors:
                out["vT_proj"] = vT
                ed.x_space_guidance_scale, ed.x_space_guidance_num_step = 0.5, 16
                ed.x_space_guidance_edit_step = 1
                xts = {}
                for direction in [1, -1]:
                    vk = direction * vT[0, :].view(-1, *x.shape[1:])
                    lst = [x.clone()]
                    for _ in range(16):
                        lst.append(ed.x_space_guidance_direct(lst[-1], t_idx=ed.edit_t_idx, vk=vk,
                                                              single_edit_step=1))
                    b = torch.cat(lst, dim=0)
                    b = b[::(b.size(0) // 2)]
                    xts[direction] = b
                xb = torch.cat([(xts[-1].flip(dims=[0]))[:-1], xts[1]], dim=0)
                check(f"{tag}/edit_batch", orc.OracleEdit.edit_batch(x, vT[0], 0.5, 16, 2), xb)
                out["edit_batch"] = xb

    # --- fixture family 6: pipeline (inversion -> xt -> eta=0 decode)
    if pipeline:
        g0 = torch.Generator().manual_seed(0)
        x0 = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=g0).clamp(-1, 1)
        ed.dataset = {0: x0}
        with torch.no_grad():
            xT = ed.run_DDIMinversion(idx=0)
            xt_e, t_e, i_e = ed.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=ed.edit_t_idx)
            ed.performance_boosting_t_idx = 1000  # eta = 0 everywhere: deterministic decode
            xdec = ed.DDIMforwardsteps(xt_e, t_start_idx=ed.edit_t_idx, t_end_idx=-1,
                                       performance_boosting=True, save_image=False)
        oxT = oed.ddim_inversion(x0)
        check(f"{tag}/inversion", oxT, xT, rtol=1e-3, atol=1e-4)
        oxt, _, oi = oed.ddim_forwardsteps(oxT, 0, oed.edit_t_idx)
        assert oi == int(i_e)
        check(f"{tag}/to_t", oxt, xt_e, rtol=1e-3, atol=1e-4)
        out["pipe_x0"], out["pipe_xT"], out["pipe_xt"], out["pipe_dec"] = x0, xT, xt_e, xdec
    return out


def gen_eta1_decode(redit, YHS, PullBackDDPM, tmpdir):
    """Fixture family 6, stochastic half: the reference's DDIMforwardsteps from the edit step to x0 with
    performance_boosting=True (eta switches 0 -> 1 at index 79, edit.py:2556-2559) on a batch of 2, with the
    `torch.randn_like` draws of YHCustomScheduler.step (utils.py:374) replaced by a recorded sequence."""
    import loco_oracle as orc
    from loco_edit_amd.config import TINY_DDPM as cfg, synth_params
    params = synth_params(cfg, seed=0)
    model = ref_model(PullBackDDPM, cfg, params)
    ed = ref_edit(redit, YHS, model, tmpdir)
    base = torch.load(os.path.join(GOLD, "tiny.pt"))
    xt = torch.cat([base["pipe_xt"], base["pipe_xt"] + 0.25 * base["x"]], dim=0)
    gn = torch.Generator().manual_seed(23)
    noises = []
    real = torch.randn_like

    def fake_randn_like(x, **kw):
        nz = torch.randn(x.shape, generator=gn)
        noises.append(nz)
        return nz

    torch.randn_like = fake_randn_like
    try:
        with torch.no_grad():
            dec = ed.DDIMforwardsteps(xt, t_start_idx=ed.edit_t_idx, t_end_idx=-1, performance_boosting=True,
                                      save_image=False)
    finally:
        torch.randn_like = real
    pbi = int(ed.performance_boosting_t_idx)
    assert len(noises) == 99 - pbi, (len(noises), pbi)
    # the oracle on the same draws
    oed = orc.OracleEdit(orc.to_torch(params), cfg)
    oed.performance_boosting_t_idx = pbi
    odec = oed.ddim_forwardsteps(xt, int(ed.edit_t_idx), -1, performance_boosting=True,
                                 noises={pbi + j: nz for j, nz in enumerate(noises)})
    check("tiny/eta1 decode", odec, dec, rtol=1e-3, atol=1e-4)
    return {"xt": xt, "pbt_idx": pbi, "first_noise_step": pbi, "noises": torch.stack(noises), "dec": dec}


def gen_scheduler(YHS, extract):
    """Fixture family 1."""
    import loco_oracle as orc
    s = YHS(Namespace(noise_schedule="linear", device=torch.device("cpu"), dtype=torch.float32))
    o = orc.Scheduler()
    check("alphas_cumprod", o.alphas_cumprod, s.alphas_cumprod, rtol=0, atol=0)
    out = {"alphas_cumprod": s.alphas_cumprod.clone()}
    s.set_timesteps(100, device="cpu")
    o.set_timesteps(100)
    check("timesteps", o.timesteps, s.timesteps, rtol=0, atol=0)
    check("timesteps_next", o.timesteps_next, s.timesteps_next, rtol=0, atol=0)
    out["fwd_timesteps"], out["fwd_timesteps_next"] = s.timesteps.clone(), s.timesteps_next.clone()
    torch.manual_seed(0)
    xt = torch.randn(1, 3, 4, 4)
    et = torch.randn(1, 3, 4, 4)
    t = s.timesteps[40]
    r = s.step(et, t, xt, eta=0)
    on, op = o.step(et, t, xt, eta=0)
    check("step eta0 prev", on, r.prev_sample, rtol=0, atol=0)
    check("step eta0 x0", op, r.x0, rtol=0, atol=0)
    out["step_xt"], out["step_et"], out["step_t"] = xt, et, t.clone()
    out["step_prev_eta0"], out["step_x0"] = r.prev_sample, r.x0
    t2 = s.timesteps[85]
    torch.manual_seed(5)
    r1 = s.step(et, t2, xt, eta=1)
    torch.manual_seed(5)
    nz = torch.randn_like(xt)
    on1, _ = o.step(et, t2, xt, eta=1, noise=nz)
    check("step eta1", on1, r1.prev_sample, rtol=0, atol=0)
    out["step_t_eta1"], out["step_noise"], out["step_prev_eta1"] = t2.clone(), nz, r1.prev_sample
    s.set_timesteps(100, device="cpu", is_inversion=True)
    o.set_timesteps(100, is_inversion=True)
    check("inv timesteps", o.timesteps, s.timesteps, rtol=0, atol=0)
    check("inv timesteps_next", o.timesteps_next, s.timesteps_next, rtol=0, atol=0)
    out["inv_timesteps"], out["inv_timesteps_next"] = s.timesteps.clone(), s.timesteps_next.clone()
    return out


def gen_celeba256_null(redit, YHS, PullBackDDPM, tmpdir, n_iter=3, k_null=5):
    """BASELINE config 2 at its size: the null-space solve of the reference on the COMPLEMENT of the l_eye-sized mask
    (edit.py:2296-2310: `mask=~mask`, L = 194 208), 256x256 CelebA-DDPM architecture, V0 injected, `n_iter` iterations
    (minutes of CPU per iteration), then the reference's projection + normalisation (edit.py:2317-2323) of the
    12-iteration modify basis of celeba256.pt (its fp16 copy, widened) against that null basis."""
    from loco_edit_amd.config import CELEBA_DDPM as cfg, synth_params
    base = torch.load(os.path.join(GOLD, "celeba256.pt"))
    model = ref_model(PullBackDDPM, cfg, synth_params(cfg, seed=0))
    ed = ref_edit(redit, YHS, model, tmpdir)
    x, t, mask = base["x"], base["t"], base["mask"]
    gv = torch.Generator().manual_seed(7)
    v0 = torch.randn(cfg.n, max(k_null, 5), generator=gv)       # the same draw the modify solve used (v0_seed 7)
    real_randn = torch.randn

    def fake_randn(*size, **kw):
        if len(size) == 2 and size[0] == cfg.n:
            return v0[:, :size[1]].clone()
        return real_randn(*size, **kw)

    torch.randn = fake_randn
    try:
        with torch.no_grad():
            u_n, s_n, vT_n = ed.local_encoder_decoder_pullback_xt(
                x=x, t=t, pca_rank=k_null, min_iter=n_iter, max_iter=n_iter, convergence_threshold=1e-4, mask=~mask)
    finally:
        torch.randn = real_randn
    vT_m = base["vT_modify_f16"].float()
    vT_null = vT_n[:k_null, :]
    vT = (vT_null.T @ (vT_null @ vT_m.T)).T
    vT = vT_m - vT
    vT = vT / vT.norm(dim=1, keepdim=True)
    gp = torch.Generator().manual_seed(17)
    P = torch.randn(cfg.n, 64, generator=gp)
    return {"n_iter": n_iter, "k_null": k_null, "v0_seed": 7, "s_null": s_n, "vT_null_f16": vT_n.to(torch.float16),
            "vT_null_proj": vT_n @ P, "vT_proj_seed": 17, "vT_projected_proj": vT @ P,
            "vT_projected_f16": vT.to(torch.float16), "u_null_norms": u_n.norm(dim=0)}


def gen_converge(redit, YHS, PullBackDDPM, tmpdir):
    """The FREE-RUNNING stop rule of edit.py:2489-2492 with the shipped arguments of run_edit_null_space_projection
    (edit.py:2292-2310: min_iter=10, max_iter=50, convergence_threshold=1e-4; scripts: --pca_rank 1 --pca_rank_null 5).

    torch.linalg.svd is wrapped to record every iterate the reference sees, so the fixture also holds, per iteration,
    which rows LAPACK returned with the opposite sign of their predecessor (`flips`) and the largest elementwise change
    up to sign (`max_delta`): with one probe the sign is stable and the loop stops when max_delta < 1e-4 (+ rtol); with
    five probes some row flips in every iteration and the loop runs all 50.  `lapack_sign_trials`: the same observation
    on random nearly diagonal k x n problems (k, n, trials, trials with a flipped first row, trials with any flipped row)."""
    import io
    import contextlib
    import loco_oracle as orc
    from loco_edit_amd.config import TINY_DDPM, MID_DDPM, synth_params
    out = {"min_iter": 10, "max_iter": 50, "convergence_threshold": 1e-4, "weights_seed": 0, "v0_seed": 7, "x_seed": 1}

    def run(cfg, k, mask, x, t, ed, v0):
        real_randn, real_svd = torch.randn, torch.linalg.svd
        vs = []

        def fake_randn(*size, **kw):
            if len(size) == 2 and size[0] == cfg.n:
                return v0[:, :size[1]].clone()
            return real_randn(*size, **kw)

        def rec_svd(A, **kw):
            r = real_svd(A, **kw)
            vs.append(r[2].clone())
            return r

        torch.randn, torch.linalg.svd = fake_randn, rec_svd
        try:
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                u, s, vT = ed.local_encoder_decoder_pullback_xt(x=x, t=t, pca_rank=k, min_iter=10, max_iter=50,
                                                                convergence_threshold=1e-4, mask=mask)
        finally:
            torch.randn, torch.linalg.svd = real_randn, real_svd
        flips = torch.stack([((vs[i - 1] * vs[i]).sum(dim=1) < 0) for i in range(1, len(vs))])
        sg = [(vs[i - 1] * vs[i]).sum(dim=1, keepdim=True).sign() for i in range(1, len(vs))]
        max_delta = torch.tensor([(vs[i - 1] - vs[i] * sg[i - 1]).abs().max().item() for i in range(1, len(vs))])
        return u, s, vT, len(vs), flips, max_delta

    for tag, cfg, mrect in (("tiny", TINY_DDPM, (15, 16, 10, 11)), ("mid", MID_DDPM, (40, 42, 12, 14))):
        params = synth_params(cfg, seed=0)
        model = ref_model(PullBackDDPM, cfg, params)
        ed = ref_edit(redit, YHS, model, tmpdir)
        oed = orc.OracleEdit(orc.to_torch(params), cfg)
        x = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(1))
        t = ed.scheduler.timesteps[int(ed.edit_t_idx)]
        mask = rect_mask(cfg, *mrect)
        v0 = torch.randn(cfg.n, 5, generator=torch.Generator().manual_seed(7))
        o = {"mask": mask, "x": x, "t": t.clone()}
        u, s, vT, n_it, flips, md = run(cfg, 1, mask, x, t, ed, v0)
        print(f"  {tag}: modify space, 1 probe: reference stopped after {n_it} iterations; rows flipped in "
              f"{int(flips.any(dim=1).sum())} of {n_it - 1} iterations; max_delta around the stop "
              f"{md[n_it - 3].item():.3e} -> {md[n_it - 2].item():.3e}")
        assert 13 <= n_it <= 40, "pick a mask whose one-probe solve stops inside the window"
        o.update(n_iter_modify=n_it, s_modify=s, vT_modify=vT, u_modify=u, flips_modify=flips, max_delta_modify=md)
        _, os_, ovT, on = oed.pullback(x, t, 1, v0[:, :1], min_iter=10, max_iter=50, convergence_threshold=1e-4, mask=mask)
        assert on == n_it, (on, n_it)
        check(f"converge/{tag}/modify s", os_, s, rtol=1e-4)
        assert abs_cos_rows(ovT, vT).min() > 0.9999
        if tag == "tiny":           # the five-probe null-space solve on the complement: every iteration has a flipped row
            u, s, vT, n_it, flips, md = run(cfg, 5, ~mask, x, t, ed, v0)
            print(f"  {tag}: null space, 5 probes: reference ran {n_it} iterations; rows flipped in "
                  f"{int(flips.any(dim=1).sum())} of {n_it - 1} iterations")
            assert n_it == 50 and bool(flips[10:].any(dim=1).all())
            o.update(n_iter_null=n_it, s_null=s, vT_null=vT, flips_null=flips, max_delta_null=md)
        out[tag] = o
    g = torch.Generator().manual_seed(0)
    trials = []
    for k in (2, 3, 5, 8, 16, 20, 64):
        for n in (3072, 12288):
            neg0 = anyflip = 0
            for _ in range(40):
                V = torch.linalg.qr(torch.randn(n, k, generator=g))[0].T
                sv = torch.sort(torch.rand(k, generator=g) * 3 + 0.5, descending=True)[0]
                A = ((torch.eye(k) + 0.02 * torch.randn(k, k, generator=g)) * sv[None, :]) @ V
                d = (torch.linalg.svd(A, full_matrices=False)[2] * V).sum(1)
                neg0 += int(d[0] < 0)
                anyflip += int((d < 0).any())
            trials.append((k, n, 40, neg0, anyflip))
            print(f"  LAPACK on nearly diagonal {k} x {n}: first row flipped {neg0}/40, any row flipped {anyflip}/40")
    out["lapack_sign_trials"] = torch.tensor(trials)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the 256x256 CelebA-DDPM summaries (minutes of CPU)")
    ap.add_argument("--full-iters", type=int, default=2)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--iters", type=int, default=12,
                    help="power iterations of the at-size solver families p2_solver / celeba256_null (the "
                         "reference's minimum is 12, edit.py:2492 with min_iter=10)")
    a = ap.parse_args()
    import tempfile
    tmpdir = tempfile.mkdtemp(prefix="loco_golden_")
    redit, YHS, extract, PullBackDDPM = import_reference()
    from loco_edit_amd.config import TINY_DDPM, MID_DDPM, CELEBA_DDPM, TINY_ADM, FFHQ_P2
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if not a.only or a.only == "sched":
        print("scheduler KATs")
        torch.save(gen_scheduler(YHS, extract), os.path.join(GOLD, "scheduler.pt"))
    if not a.only or a.only == "tiny":
        print("tiny config (32x32, ch 32)")
        o = gen_for_config("tiny", TINY_DDPM, redit, YHS, PullBackDDPM, k=5, k_null=5, n_iter=12,
                           mrect=(12, 20, 8, 18), tmpdir=tmpdir)
        torch.save(o, os.path.join(GOLD, "tiny.pt"))
    if not a.only or a.only == "mid":
        print("mid config (64x64, ch 32)")
        o = gen_for_config("mid", MID_DDPM, redit, YHS, PullBackDDPM, k=3, k_null=0, n_iter=3,
                           mrect=(20, 36, 10, 40), tmpdir=tmpdir, pipeline=False)
        torch.save(o, os.path.join(GOLD, "mid.pt"))
    if not a.only or a.only == "tiny_adm":
        print("tiny ADM/P2 config (32x32, ch 32, 4 heads)")
        o = gen_for_config("tiny_adm", TINY_ADM, redit, YHS, PullBackDDPM, k=4, k_null=0, n_iter=12,
                           mrect=(12, 20, 8, 18), tmpdir=tmpdir, pipeline=False)
        torch.save(o, os.path.join(GOLD, "tiny_adm.pt"))
    if not a.only or a.only == "tiny_adm_plain":
        print("tiny guided-diffusion config without scale-shift norm, conv down / up-sampling (the latent-diffusion skeleton)")
        from loco_edit_amd.config import TINY_ADM_PLAIN
        o = gen_for_config("tiny_adm_plain", TINY_ADM_PLAIN, redit, YHS, PullBackDDPM, k=4, k_null=0, n_iter=3,
                           mrect=(12, 20, 8, 18), tmpdir=tmpdir, pipeline=False)
        torch.save(o, os.path.join(GOLD, "tiny_adm_plain.pt"))
    if a.only == "p2_256":
        print("full FFHQ-P2 config (256x256)")
        o = gen_for_config("p2_256", FFHQ_P2, redit, YHS, PullBackDDPM, k=4, k_null=0, n_iter=0,
                           mrect=(110, 130, 70, 110), tmpdir=tmpdir, full_tensors=False, pipeline=False)
        o = {k: v for k, v in o.items() if v is not None}
        torch.save(o, os.path.join(GOLD, "p2_256.pt"))
    if a.only == "p2_solver":
        print(f"FFHQ-P2 256x256 solver fixture: 16 probes, {a.iters} iterations")
        o = gen_for_config("p2_solver", FFHQ_P2, redit, YHS, PullBackDDPM, k=16, k_null=0, n_iter=a.iters,
                           mrect=(110, 130, 70, 110), tmpdir=tmpdir, full_tensors=False, pipeline=False,
                           oracle_solver=False)
        keep = ("cfg", "weights_seed", "x", "t", "mask", "v0_seed", "n_iter", "s_modify", "vT_modify_f16",
                "vT_proj_seed", "vT_modify_proj")
        torch.save({k: o[k] for k in keep}, os.path.join(GOLD, "p2_solver.pt"))
    if a.only == "celeba256_null":
        print(f"config 2 at size: null-space solve on the complement mask, {a.iters} iterations + projection")
        torch.save(gen_celeba256_null(redit, YHS, PullBackDDPM, tmpdir, n_iter=a.iters),
                   os.path.join(GOLD, "celeba256_null.pt"))
    if a.only == "converge":
        print("free-running stop rule (shipped arguments) on the tiny and mid configs")
        torch.save(gen_converge(redit, YHS, PullBackDDPM, tmpdir), os.path.join(GOLD, "converge.pt"))
    if a.only == "eta1":
        print("tiny eta=1 decode with injected noise")
        torch.save(gen_eta1_decode(redit, YHS, PullBackDDPM, tmpdir), os.path.join(GOLD, "tiny_eta1.pt"))
    if a.full:
        print("full config (256x256 CelebA-HQ DDPM arch)")
        o = gen_for_config("celeba256", CELEBA_DDPM, redit, YHS, PullBackDDPM, k=5, k_null=0,
                           n_iter=a.full_iters, mrect=(110, 130, 70, 110), tmpdir=tmpdir,
                           full_tensors=False, pipeline=False)
        o = {k: v for k, v in o.items() if v is not None}
        torch.save(o, os.path.join(GOLD, "celeba256.pt"))
    print("done ->", GOLD)


if __name__ == "__main__":
    main()
