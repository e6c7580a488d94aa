"""MI355X parity tests: the HIP path (through the C ABI) against the golden
vectors captured from the reference and against the CPU oracle on the same
seeded inputs, in BOTH conv arithmetic modes.  Stated tolerances:
  f32    (exact fp32 MFMA):      single pass rel-L2 <= 2e-5
  bf16x3 (split-bf16, default):  single pass rel-L2 <= 1e-4 (measured ~1.4e-5)
  solver after 12 iterations: |cos(vT_i)| >= 0.9999 (0.999 at 256^2), s rtol 1e-3
  deterministic decode: PSNR >= 60 dB (f32) / 35 dB (bf16x3; chaotic 138-step chain of the untrained net, per-step rel-L2 <= 1e-4)
(north_star bar: |cos| >= 0.99)."""
import math
import os

import pytest
import torch

import loco_oracle as orc
from loco_edit_amd.config import CELEBA_DDPM, FFHQ_P2, MID_DDPM, TINY_ADM, TINY_DDPM, UNetConfig, synth_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def psnr(a, b, peak=2.0):
    mse = ((a.detach().cpu().double() - b.detach().cpu().double()) ** 2).mean().item()
    return 10 * math.log10(peak * peak / max(mse, 1e-30))


PRECS = ["f32", "bf16x3"]
TOL = {"f32": 2e-5, "bf16x3": 1e-4}


@pytest.fixture(scope="module")
def engines():
    from loco_edit_amd.hip import LocoEngine, library_path
    assert os.path.exists(library_path())
    cache = {}

    def get(cfg, prec="f32"):
        if cfg not in cache:
            e = LocoEngine(cfg, max_batch=8, device=torch.device(DEV))
            e.load_state_dict(synth_params(cfg, 0))
            cache[cfg] = e
        cache[cfg].set_precision(prec)
        return cache[cfg]
    return get


def _sched():
    s = orc.Scheduler()
    s.set_timesteps(100)
    return s


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag,cfg", [("tiny", TINY_DDPM), ("mid", MID_DDPM), ("tiny_adm", TINY_ADM)])
def test_forward_jvp_vjp_vs_golden(tag, cfg, prec, engines, golden):
    g = golden(tag)
    eng = engines(cfg, prec)
    tol = TOL[prec]
    x, t = g["x"].to(DEV), float(g["t"])
    eps = eng.unet_forward(x, t)
    assert rel(eps, g["eps"]) < tol
    # batch of identical images at once == single (batch stride handling)
    eps3 = eng.unet_forward(x.repeat(3, 1, 1, 1).contiguous(), t)
    assert torch.equal(eps3[2], eps3[0]) and rel(eps3[1:2], g["eps"]) < tol
    at = float(_sched().alpha_at(g["t"]))
    eng.pmp_primal(x, t, at, g["mask"].to(DEV))
    k = g["V"].shape[0]
    U = eng.pmp_jvp(g["V"].reshape(k, -1).contiguous().to(DEV))
    assert rel(eng.mask_gather(U), g["JV"]) < tol
    assert float(U[:, ~g["mask"].reshape(-1).to(DEV)].abs().max()) == 0.0
    Uin = torch.zeros(k, cfg.n)
    Uin[:, g["mask"].reshape(-1)] = g["JV"]
    A = eng.pmp_vjp(Uin.to(DEV))
    assert rel(A, g["UtJ"]) < tol


@pytest.mark.parametrize("prec", PRECS)
def test_unmasked_and_et_operators(prec, engines):
    cfg = TINY_DDPM
    eng = engines(cfg, prec)
    tol = TOL[prec]
    oed = orc.OracleEdit(orc.to_torch(synth_params(cfg, 0)), cfg)
    s = _sched()
    t = s.timesteps[40]
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    V = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    for noise in (False, True):
        eng.pmp_primal(x.to(DEV), float(t), float(s.alpha_at(t)), None, use_et=noise)
        U = eng.pmp_jvp(V.reshape(2, -1).contiguous().to(DEV))
        Uo = orc.jvp_x0(oed, x, t, V, mask=None, noise=noise)
        assert rel(U, Uo.reshape(2, -1)) < tol
        A = eng.pmp_vjp(U)
        Ao = orc.vjp_x0(oed, x, t, Uo, mask=None, noise=noise)
        assert rel(A, Ao) < 1.5 * tol


@pytest.mark.parametrize("prec", PRECS)
def test_adjointness_and_linearity_full_size(prec, engines):
    """Size-independent properties at BASELINE.json's full 256x256 size."""
    cfg = CELEBA_DDPM
    eng = engines(cfg, prec)
    s = _sched()
    t = s.timesteps[40]
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(DEV)
    mask = torch.zeros(3, 256, 256, dtype=torch.bool)
    mask[:, 110:130, 70:110] = True
    eng.pmp_primal(x, float(t), float(s.alpha_at(t)), mask.to(DEV))
    V = torch.randn(3, cfg.n, generator=torch.Generator().manual_seed(5)).to(DEV)
    U = torch.randn(3, cfg.n, generator=torch.Generator().manual_seed(6)).to(DEV) * mask.reshape(1, -1).to(DEV)
    JV = eng.pmp_jvp(V)
    JtU = eng.pmp_vjp(U)
    lhs, rhs = (JV * U).sum(dim=1), (V * JtU).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1))).max().item() < 1e-4
    comb = (2.0 * V[0] - 0.5 * V[1])[None].contiguous()
    assert rel(eng.pmp_jvp(comb)[0], 2.0 * JV[0] - 0.5 * JV[1]) < 1e-4


@pytest.mark.parametrize("prec", PRECS)
def test_full_size_forward_vs_golden_samples(prec, engines, golden):
    path = os.path.join(os.path.dirname(__file__), "golden", "celeba256.pt")
    if not os.path.exists(path):
        pytest.skip("256x256 summaries not generated")
    g = golden("celeba256")
    eng = engines(CELEBA_DDPM, prec)
    tol = TOL[prec]
    eps = eng.unet_forward(g["x"].to(DEV), float(g["t"]))
    assert rel(eps.reshape(-1)[g["eps_sample_idx"].to(DEV)], g["eps_sample"]) < tol
    assert abs(eps.double().sum().item() - g["eps_sum"]) < 1e-3 * math.sqrt(g["eps_sqsum"])
    at = float(_sched().alpha_at(g["t"]))
    eng.pmp_primal(g["x"].to(DEV), float(g["t"]), at, g["mask"].to(DEV))
    k = g["JV"].shape[0]
    v0 = torch.randn(CELEBA_DDPM.n, k, generator=torch.Generator().manual_seed(g["v0_seed"]))
    V = torch.linalg.qr(v0)[0].T.contiguous()
    U = eng.pmp_jvp(V.to(DEV))
    assert rel(eng.mask_gather(U), g["JV"]) < 2.5 * tol
    Uin = torch.zeros(k, CELEBA_DDPM.n)
    Uin[:, g["mask"].reshape(-1)] = g["JV"]
    A = eng.pmp_vjp(Uin.to(DEV)).cpu()
    P = torch.randn(CELEBA_DDPM.n, 64, generator=torch.Generator().manual_seed(g["UtJ_proj_seed"]))
    assert rel(A @ P, g["UtJ_proj"]) < 2.5 * tol
    assert torch.allclose(A.norm(dim=1), g["UtJ_norm"], rtol=2e-4)
    if "s_modify" in g:
        from loco_edit_amd import solver
        u, s, vT, n_it = solver.local_basis(eng, g["x"].to(DEV), float(g["t"]), at, k, mask=g["mask"].to(DEV),
                                            min_iter=g["n_iter"], max_iter=g["n_iter"], v0=v0.to(DEV), verbose=False)
        assert torch.allclose(s.cpu(), g["s_modify"], rtol=1e-3)
        cos = (vT.cpu() * g["vT_modify_f16"].float()).sum(dim=1).abs()
        assert cos.min().item() > 0.999


def test_p2_full_size_vs_reference_golden(engines, golden):
    """Denoiser B (FFHQ-P2 / guided-diffusion U-Net, BASELINE.json config 2) at 256x256 against the
    reference's own UNetModel outputs (sampled eps, J V on the mask, projections of U^T J)."""
    g = golden("p2_256")
    eng = engines(FFHQ_P2, "bf16x3")
    tol = TOL["bf16x3"]
    eps = eng.unet_forward(g["x"].to(DEV), float(g["t"]))
    assert rel(eps.reshape(-1)[g["eps_sample_idx"].to(DEV)], g["eps_sample"]) < tol
    assert abs(eps.double().sum().item() - g["eps_sum"]) < 1e-3 * math.sqrt(g["eps_sqsum"])
    at = float(_sched().alpha_at(g["t"]))
    eng.pmp_primal(g["x"].to(DEV), float(g["t"]), at, g["mask"].to(DEV))
    k = g["JV"].shape[0]
    v0 = torch.randn(FFHQ_P2.n, k, generator=torch.Generator().manual_seed(g["v0_seed"]))
    V = torch.linalg.qr(v0)[0].T.contiguous()
    U = eng.pmp_jvp(V.to(DEV))
    assert rel(eng.mask_gather(U), g["JV"]) < 2.5 * tol
    Uin = torch.zeros(k, FFHQ_P2.n)
    Uin[:, g["mask"].reshape(-1)] = g["JV"]
    A = eng.pmp_vjp(Uin.to(DEV)).cpu()
    P = torch.randn(FFHQ_P2.n, 64, generator=torch.Generator().manual_seed(g["UtJ_proj_seed"]))
    assert rel(A @ P, g["UtJ_proj"]) < 2.5 * tol


def test_adm_solver_and_many_probes(engines, golden):
    """tiny P2-style model: 12-iteration solver vs the reference golden, then the config-2 idiom
    'rank-r basis from k >> r probes' (pca_rank=k, keep vT[:r]; edit.py:2320 slicing)."""
    from loco_edit_amd import solver
    g = golden("tiny_adm")
    eng = engines(TINY_ADM, "bf16x3")
    at = float(_sched().alpha_at(g["t"]))
    x = g["x"].to(DEV)
    u, s, vT, n_it = solver.local_basis(eng, x, float(g["t"]), at, 4, mask=g["mask"].to(DEV), min_iter=g["n_iter"],
                                        max_iter=g["n_iter"], v0=g["v0"].to(DEV), verbose=False)
    assert torch.allclose(s.cpu(), g["s_modify"], rtol=1e-3)
    assert (vT.cpu() * g["vT_modify"]).sum(dim=1).abs().min().item() > 0.999
    # 16 probes (two chunks of max_batch=8), keep the top 4: the leading subspace agrees with the 4-probe run
    v0 = torch.randn(TINY_ADM.n, 16, generator=torch.Generator().manual_seed(9)).to(DEV)
    u2, s2, vT2, _ = solver.local_basis(eng, x, float(g["t"]), at, 16, mask=g["mask"].to(DEV), min_iter=12,
                                        max_iter=12, v0=v0, verbose=False)
    assert vT2.shape == (16, TINY_ADM.n) and bool((s2[:-1] >= s2[1:] - 1e-4).all())
    assert torch.allclose(s2[:2].cpu(), g["s_modify"][:2], rtol=2e-2)
    overlap = torch.linalg.svdvals(vT2[:4].cpu() @ g["vT_modify"][:2].T)
    assert overlap.min().item() > 0.98


@pytest.mark.parametrize("prec", PRECS)
def test_solver_vs_reference_golden(prec, engines, golden):
    from loco_edit_amd import solver
    g = golden("tiny")
    eng = engines(TINY_DDPM, prec)
    at = float(_sched().alpha_at(g["t"]))
    x = g["x"].to(DEV)
    u, s, vT, n_it = solver.local_basis(eng, x, float(g["t"]), at, 5, mask=g["mask"].to(DEV),
                                        min_iter=g["n_iter"], max_iter=g["n_iter"], convergence_threshold=1e-4,
                                        v0=g["v0"].to(DEV), verbose=False)
    assert n_it == g["n_iter"]
    assert torch.allclose(s.cpu(), g["s_modify"], rtol=1e-3)
    cos = (vT.cpu() * g["vT_modify"]).sum(dim=1).abs()
    assert cos.min().item() > 0.9999, cos
    ucos = torch.nn.functional.cosine_similarity(u.cpu().T, g["u_modify"].T, dim=1).abs()
    assert ucos.min().item() > 0.999
    assert (vT @ vT.T - torch.eye(5, device=DEV)).abs().max().item() < 1e-5
    # null-space solve on the complement mask + projection (edit.py:2307-2323)
    un, sn, vTn, _ = solver.local_basis(eng, x, float(g["t"]), at, 5, mask=(~g["mask"]).to(DEV),
                                        min_iter=g["n_iter"], max_iter=g["n_iter"], v0=g["v0"].to(DEV), verbose=False)
    assert torch.allclose(sn.cpu(), g["s_null"], rtol=1e-3)
    assert (vTn.cpu() * g["vT_null"]).sum(dim=1).abs().min().item() > 0.999
    proj = eng.null_project(g["vT_modify"].to(DEV).contiguous(), g["vT_null"].to(DEV).contiguous())
    assert rel(proj, g["vT_proj"]) < 1e-5
    only_norm = eng.null_project(g["vT_modify"].to(DEV).contiguous() * 3.0, None)
    assert rel(only_norm, g["vT_modify"]) < 1e-5


def test_solver_algebra_kernels(engines):
    eng = engines(TINY_DDPM)
    n = TINY_DDPM.n
    for k in (1, 5, 16, 64):
        A0 = torch.randn(k, n, generator=torch.Generator().manual_seed(k)) * torch.linspace(3, 0.5, k)[:, None]
        A = A0.to(DEV).clone()
        s = eng.orthonormalize_(A)
        _, st, vt = torch.linalg.svd(A0.double(), full_matrices=False)
        assert torch.allclose(s.cpu().double(), st, rtol=1e-4)
        assert (A @ A.T - torch.eye(k, device=DEV)).abs().max().item() < 2e-5
        if k <= 16:
            assert (A.cpu().double() * vt).sum(dim=1).abs().min().item() > 0.999
        Q = A0.to(DEV).clone()
        eng.qr_rows_(Q)
        assert (Q @ Q.T - torch.eye(k, device=DEV)).abs().max().item() < 2e-5
        qt = torch.linalg.qr(A0.double().T)[0].T
        assert (Q.cpu().double() * qt).sum(dim=1).abs().min().item() > 0.9999
    a = torch.randn(5, n).to(DEV)
    b = a + 5e-4
    out = eng.convergence(a, b, 1e-3).tolist()
    assert abs(out[0] - 5e-4 * math.sqrt(5 * n)) / out[0] < 1e-3 and out[1] == 1.0
    assert eng.convergence(a, b, 1e-4).tolist()[1] == 0.0


def test_scheduler_step_and_edit_kernels(engines, golden):
    from loco_edit_amd.scheduler import YHCustomScheduler
    g = golden("scheduler")
    eng = engines(TINY_DDPM)
    s = YHCustomScheduler(engine=eng)
    s.set_timesteps(100)
    out = s.step(g["step_et"].to(DEV), g["step_t"], g["step_xt"].to(DEV), eta=0)
    assert torch.allclose(out.prev_sample.cpu(), g["step_prev_eta0"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(out.x0.cpu(), g["step_x0"], rtol=1e-6, atol=1e-5)
    out1 = s.step(g["step_et"].to(DEV), g["step_t_eta1"], g["step_xt"].to(DEV), eta=1, noise=g["step_noise"].to(DEV))
    assert torch.allclose(out1.prev_sample.cpu(), g["step_prev_eta1"], rtol=1e-6, atol=1e-6)
    gt = golden("tiny")
    xb = eng.edit_axpy(gt["x"].to(DEV), gt["vT_proj"][0].to(DEV).contiguous(), [-8.0, -4.0, 0.0, 4.0, 8.0])
    assert torch.allclose(xb.cpu(), gt["edit_batch"], rtol=1e-5, atol=1e-5)


def _edit_obj(eng, cfg, tmp_path, **kw):
    os.environ["LOCO_PRECISION"] = kw.get("prec", "bf16x3")
    from argparse import Namespace
    from loco_edit_amd.edit import EditUncondDiffusion
    import loco_edit_amd.utils as lu
    args = Namespace(device=torch.device(DEV), dtype=torch.float32, seed=1, model_name="tiny", unet_config=cfg,
                     synthetic_weights=0, ckpt_path="", max_batch=8, image_size=cfg.resolution, c_in=3,
                     dataset_name="Synthetic", dataset_root="", for_steps=100, inv_steps=100,
                     use_yh_custom_scheduler=True, edit_t=0.6, performance_boosting_t=kw.get("pbt", 0.2),
                     x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.5, x_space_guidance_num_step=16,
                     result_folder=str(tmp_path), sample_idx=0, vT_path=kw.get("vT_path", ""), vT1_path="",
                     choose_sem="l_eye", mask_index=0, sampling_mode=False)
    return EditUncondDiffusion(args)


@pytest.mark.parametrize("prec", PRECS)
def test_pipeline_vs_reference_golden(prec, engines, golden, tmp_path, capsys):
    """inversion -> x_t -> eta=0 decode (fixture family 6) through the reference-shaped class."""
    g = golden("tiny")
    ed = _edit_obj(None, TINY_DDPM, tmp_path, pbt=0.0, prec=prec)
    assert ed.engine.get_precision() == prec
    # The untrained synthetic denoiser is not contractive: over the 138-step inversion + sampling chain a
    # per-step perturbation is amplified ~10^3x (values reach +-34), so the chained bound for the 2^-16-faithful
    # mode is looser; the per-step bar is checked separately below.
    floor = 60 if prec == "f32" else 35
    s1 = ed.scheduler
    s1.set_timesteps(100, is_inversion=True)
    t0 = s1.timesteps[50]
    one = ed._step(g["pipe_xT"].to(DEV), t0, eta=0)
    so = orc.Scheduler(); so.set_timesteps(100, is_inversion=True)
    oed1 = orc.OracleEdit(orc.to_torch(synth_params(TINY_DDPM, 0)), TINY_DDPM)
    with torch.no_grad():
        ref1, _ = so.step(oed1.unet(g["pipe_xT"], so.timesteps[50]), so.timesteps[50], g["pipe_xT"], eta=0)
    assert rel(one, ref1) < TOL[prec]
    assert ed.edit_t_idx == 40 and ed.performance_boosting_t_idx == 1000
    xT = ed.run_DDIMinversion(idx=0, x0=g["pipe_x0"])
    assert psnr(xT, g["pipe_xT"], peak=8.0) > floor
    xt, t, i = ed.DDIMforwardsteps(xT, t_start_idx=0, t_end_idx=ed.edit_t_idx)
    assert i == 40 and abs(float(t) - 595.3636) < 1e-3
    assert psnr(xt, g["pipe_xt"], peak=8.0) > floor
    ed.EXP_NAME = "dec"
    dec = ed.DDIMforwardsteps(xt, t_start_idx=ed.edit_t_idx, t_end_idx=-1, performance_boosting=True)
    # the synthetic (untrained) denoiser leaves [-1,1]: PSNR against the reference's own value range
    assert psnr(dec, g["pipe_dec"], peak=float(g["pipe_dec"].max() - g["pipe_dec"].min())) > floor
    assert os.path.exists(os.path.join(ed.result_folder, "dec.png"))
    assert os.path.exists(os.path.join(ed.result_folder, "original.png"))
    # get_x0 / get_et seams
    x0m = ed.get_x0(t, g["x"].to(DEV), mask=g["mask"])
    oed = orc.OracleEdit(orc.to_torch(synth_params(TINY_DDPM, 0)), TINY_DDPM)
    with torch.no_grad():
        assert rel(x0m, oed.get_x0(g["t"], g["x"], mask=g["mask"])) < TOL[prec]


def test_run_edit_null_space_projection_end_to_end(tmp_path):
    """The reference entry point on the tiny config: files, shapes, vT_path round trip."""
    ed = _edit_obj(None, TINY_DDPM, tmp_path)
    xt = ed.run_edit_null_space_projection(idx=0, vis_num=2, vis_num_pc=1, pca_rank=1, pca_rank_null=3,
                                           null_space_projection=True, use_mask=True)
    assert xt.shape == (5, 3, 32, 32)
    bdir = os.path.join(ed.result_folder, "basis", "local_basis-0.6T-select-mask-l_eye")
    assert os.path.exists(os.path.join(bdir, "vT-modify-pca-rank-1.pt"))
    assert os.path.exists(os.path.join(bdir, "vT-null-3.pt"))
    pcs = sorted(f for f in os.listdir(bdir) if f.endswith("-vT.pt"))
    assert len(pcs) == 1 and "pc_000" in pcs[0]
    v = torch.load(os.path.join(bdir, pcs[0]))
    assert v.shape == (1, TINY_DDPM.n) and v.dtype == torch.float32
    assert abs(float(v.norm()) - 1.0) < 1e-4
    # projected direction is orthogonal to the null basis
    vn = torch.load(os.path.join(bdir, "vT-null-3.pt")).to(v.device)
    assert (vn @ v.T).abs().max().item() < 1e-4
    # --vT_path short-circuits the solver (edit.py:2333-2336)
    ed2 = _edit_obj(None, TINY_DDPM, tmp_path, vT_path=os.path.join(bdir, pcs[0]))
    xt2 = ed2.run_edit_null_space_projection(idx=0, vis_num=2, vis_num_pc=1, pca_rank=1, pca_rank_null=3,
                                             null_space_projection=True)
    assert torch.allclose(xt2, xt, rtol=1e-5, atol=1e-5)
    pngs = [f for f in os.listdir(ed.result_folder) if f.endswith(".png")]
    assert any("Edit-random" in f for f in pngs)


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    """The sharded bench path end to end (rendezvous, probe sharding, the per-iteration all-gather, barrier +
    max-over-ranks timing, the extra profiled step on every rank, one JSON line from rank 0) with two ranks on one
    GPU over gloo; the 8-GPU RCCL run itself is the driver's."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LOCO_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["probes_total"] == 10 and d["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"] is None


@pytest.mark.gpu
def test_edge_cases_empty_mask_single_probe_full_mask(engines):
    """Empty mask is rejected (the reference would produce NaN directions); a single probe and an all-true mask
    (L = n, same operator as mask=None) run through the solver."""
    from loco_edit_amd import solver
    cfg = TINY_DDPM
    eng = engines(cfg, "f32")
    s = _sched()
    t = float(s.timesteps[40]); at = float(s.alpha_at(t))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=g).to(DEV)
    empty = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool, device=DEV)
    with pytest.raises(ValueError):
        solver.local_basis(eng, x, t, at, 2, mask=empty, min_iter=2, max_iter=3, verbose=False)
    v0 = torch.randn(cfg.n, 1, generator=g)
    u, sv, vT, n_it = solver.local_basis(eng, x, t, at, 1, mask=~empty, min_iter=2, max_iter=4, v0=v0, verbose=False)
    u2, sv2, vT2, _ = solver.local_basis(eng, x, t, at, 1, mask=None, min_iter=2, max_iter=4, v0=v0, verbose=False)
    assert vT.shape == (1, cfg.n) and u.shape == (cfg.n, 1) and torch.isfinite(vT).all()
    assert abs(float(vT.norm()) - 1.0) < 1e-4
    assert float((vT * vT2).sum().abs()) > 0.99999 and abs(float(sv[0] - sv2[0])) < 1e-4 * float(sv2[0])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [5, 6])
def test_probe_batching_is_invariant_full_size(k, engines):
    """J V and J^T U of a probe do not depend on how the probes are batched (5 and 6 probes exercise the tail-probe
    split with one and two probes in the split-K tail; single-probe launches take neither path)."""
    cfg = CELEBA_DDPM
    eng = engines(cfg, "bf16x3")
    s = _sched()
    t = float(s.timesteps[40]); at = float(s.alpha_at(t))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=g).to(DEV)
    mask = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool)
    mask[:, 110:130, 70:110] = True
    eng.pmp_primal(x, t, at, mask.to(DEV))
    V = torch.randn(k, cfg.n, generator=g).to(DEV)
    U = eng.pmp_jvp(V)
    A = eng.pmp_vjp(U)
    for i in (0, k - 1):
        Ui = eng.pmp_jvp(V[i:i + 1].contiguous())
        Ai = eng.pmp_vjp(U[i:i + 1].contiguous())
        assert float((U[i] - Ui[0]).norm() / Ui[0].norm()) < 1e-5     # split-K factors differ with the batch: rounding only
        assert float((A[i] - Ai[0]).norm() / Ai[0].norm()) < 1e-5
