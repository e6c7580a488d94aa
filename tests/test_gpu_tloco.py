"""MI355X: the pixel-space T-LOCO path (loco_edit_amd.tloco.EditDeepFloydIF on the HIP engine, one engine context per
prompt, CFG-combined Jacobian) against the fixture produced by the reference's own EditDeepFloydIF methods on the same
stand-in conditional denoiser (tests/golden/tloco_tiny.pt).  Tolerances: single evaluation rel-L2 <= 2e-5 (f32) /
1e-4 (bf16x3); solver s rtol 1e-3, |cos(vT_i)| >= 0.999; directions |cos| >= 0.9999."""
import os
from argparse import Namespace

import pytest
import torch

from loco_edit_amd.config import TINY_ADM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"f32": 2e-5, "bf16x3": 1e-4}


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cosrow(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a * b).sum(dim=1) / (a.norm(dim=1) * b.norm(dim=1))).abs()


def _edit(g, tmp_path, prec, **kw):
    from loco_edit_amd.tloco import EditDeepFloydIF
    os.environ.pop("WORLD_SIZE", None)
    args = Namespace(device=torch.device(DEV), dtype=torch.float32, seed=1, unet_config=kw.get("cfg", TINY_ADM), synthetic_weights=0, ckpt_path="",
                     max_batch=8, precision=prec, dataset_name="Random", for_steps=100, use_yh_custom_scheduler=True,
                     guidance_scale=g["guidance_scale"], guidance_scale_edit=g["guidance_scale_edit"],
                     prompt_emb={"for": g["for_e"], "edit": g["edit_e"], "null": g["null_e"]}, for_prompt="a cat",
                     edit_prompt="a dog", edit_t=0.6, sampling_mode=False, tilda_v_score_type=kw.get("tilda", "null+(for-null)+(edit-null)"),
                     ablation_method=kw.get("ablation", "null-space-proj"), mask_type="SAM", vT_path=kw.get("vT_path", ""),
                     x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.5, x_space_guidance_num_step=16,
                     result_folder=str(tmp_path))
    return EditDeepFloydIF(args)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_tloco_pieces_vs_reference_golden(prec, golden, tmp_path):
    g = golden("tloco_tiny")
    ed = _edit(g, tmp_path, prec)
    tol = TOL[prec]
    x, t = g["x"].to(DEV), g["t"]
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    assert ed.edit_t_idx == g["edit_t_idx"] and float(ed.scheduler.timesteps[ed.edit_t_idx]) == float(t)
    xb = torch.cat([g["x"], g["x"].flip(-1)], dim=0).to(DEV)
    # 1. CFG noise, every mode (edit.py:1286-1373)
    for mode, ref in g["eps_modes"].items():
        assert rel(ed._classifer_free_guidance(xb, t, F, E, N, mode, True), ref) < 4 * tol, mode
    assert rel(ed._classifer_free_guidance(xb, t, F, E, N, "null+(for-null)", False), g["eps_nocfg"][:, :3]) < tol
    # the conditioning does something: the three branches differ
    ef, en = ed.branches["for"].unet_forward(x, float(t)), ed.branches["null"].unet_forward(x, float(t))
    assert rel(ef, en) > 1e-2
    # 2. get_x0 (edit.py:1566-1587)
    assert rel(ed.get_x0(x, t, ed.edit_t_idx, F, E, N, mask=g["mask"]), g["x0_masked"]) < 4 * tol
    # 3. CFG-combined subspace solver (edit.py:1589-1676), three modes / masks
    for mode, sv in g["solver"].items():
        m = None if sv["mask"] is None else sv["mask"]
        u, s, vT = ed.local_encoder_decoder_pullback_xt(x, t, ed.edit_t_idx, F, E, N, pca_rank=3, min_iter=sv["n_iter"],
                                                        max_iter=sv["n_iter"], mask=m, mode=mode, v0=g["v0"].to(DEV), verbose=False)
        assert ed.last_n_iter == sv["n_iter"]
        assert torch.allclose(s.cpu(), sv["s"], rtol=1e-3), (mode, s.cpu(), sv["s"])
        assert cosrow(vT, sv["vT"]).min().item() > 0.999, mode
        assert cosrow(u.T, sv["u"].T).min().item() > 0.999, mode
    # 4. direction through the Jacobian (edit.py:1680-1717), 5. direct directions (:1720-1741)
    vg = ed.get_delta_xt_via_grad(x, t, ed.edit_t_idx, F, E, N, mask=g["mask"], mode="null+(for-null)+(edit-null)")
    assert cosrow(vg, g["v_grad"]).item() > 0.9999 and abs(float(vg.norm()) - 1.0) < 1e-4
    sgn = torch.sign((vg.cpu() * g["v_grad"]).sum())
    assert float(sgn) == 1.0                                            # a direction, not a line: the sign is part of the edit
    for mode, ref in g["v_direct"].items():
        vd = ed.get_v_modify(x, t, ed.edit_t_idx, F, E, N, mask=g["mask"], mode=mode, jacobian=False)
        c = ((vd.cpu().double() * ref.double()).sum() / ref.double().norm()).item()
        assert c > 0.9999, (mode, c)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_tloco_sampler_vs_reference_golden(prec, golden, tmp_path):
    """DDPMforwardsteps (edit.py:1412-1481): x_T -> x_t at the edit step under 'null+(for-null)' guidance, then the
    decode of a 2-image batch under 'null+(for-null)+(edit-null)' to the uint8 image tensor."""
    g = golden("tloco_tiny")
    ed = _edit(g, tmp_path, prec)
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    xt, t, i = ed.DDPMforwardsteps(g["xT"].to(DEV), 0, ed.edit_t_idx, F, E, N, mode="null+(for-null)")
    assert i == g["edit_t_idx"] and float(t) == float(g["t_edit"])
    ref = g["xt_edit"]
    mse = ((xt.cpu().double() - ref.double()) ** 2).mean().item()
    peak = float(ref.max() - ref.min())
    import math
    assert 10 * math.log10(peak * peak / max(mse, 1e-30)) > (60 if prec == "f32" else 35)
    ed.EXP_NAME = "dec"
    img = ed.DDPMforwardsteps(g["dec_in"].to(DEV), ed.edit_t_idx, -1, F, E, N, mode="null+(for-null)+(edit-null)")
    assert img.dtype == torch.uint8 and tuple(img.shape) == tuple(g["dec_u8"].shape) == (2, 32, 32, 3)
    differ = (img.cpu().int() - g["dec_u8"].int()).abs()
    assert float((differ > 1).float().mean()) < (0.002 if prec == "f32" else 0.05)
    assert os.path.exists(os.path.join(ed.result_folder, "dec_stage1.png"))


def test_cfg_operator_adjoint_and_linear(golden, tmp_path):
    """<J V, U> == <V, J^T U> and linearity for the CFG-combined operator, mode '(for-edit)' (weights sum to 0:
    the identity part of d x0_hat / d x_t must not be scaled by the guidance weights)."""
    g = golden("tloco_tiny")
    ed = _edit(g, tmp_path, "f32")
    x, t = g["x"].to(DEV), g["t"]
    op = ed._operator(x, t, g["mask"], "(for-edit)")
    gen = torch.Generator().manual_seed(3)
    V = torch.randn(2, TINY_ADM.n, generator=gen).to(DEV)
    U = (torch.randn(2, TINY_ADM.n, generator=gen) * g["mask"].reshape(1, -1)).to(DEV)
    JV, JtU = op.jvp(V), op.vjp(U)
    lhs, rhs = (JV * U).sum(dim=1), (V * JtU).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1))).max().item() < 1e-4
    assert float(JV[:, ~g["mask"].reshape(-1).to(DEV)].abs().max()) == 0.0
    comb = (2.0 * V[0] - 0.5 * V[1])[None].contiguous()
    assert rel(op.jvp(comb)[0], 2.0 * JV[0] - 0.5 * JV[1]) < 1e-4
    # finite-difference check of J V against get_x0 in the same mode (a denoiser evaluation overwrites the primal arena:
    # the operator is rebuilt afterwards)
    h = 1e-2
    v = V[0].view(1, 3, 32, 32) / V[0].norm()
    x0p = ed.get_x0(x + h * v, t, ed.edit_t_idx, g["for_e"], g["edit_e"], g["null_e"], mask=g["mask"], mode="(for-edit)")
    x0m = ed.get_x0(x - h * v, t, ed.edit_t_idx, g["for_e"], g["edit_e"], g["null_e"], mask=g["mask"], mode="(for-edit)")
    fd = (x0p - x0m) / (2 * h)
    with pytest.raises(RuntimeError):
        op.jvp(V)                                   # stale cache is refused, not silently used
    op = ed._operator(x, t, g["mask"], "(for-edit)")
    jv = op.gather(op.jvp((V[0] / V[0].norm())[None].contiguous()))
    assert rel(jv, fd) < 2e-2


def test_tloco_drivers_end_to_end(golden, tmp_path):
    """run_edit_null_space_projection_xt (edit.py:1745-1868) and ..._xt_semantic (:1871-2018, jacobian direction,
    null-space projected): files, shapes, unit norm, orthogonality to the null basis, --vT_path reload."""
    g = golden("tloco_tiny")
    ed = _edit(g, tmp_path, "bf16x3")
    masks = torch.zeros(2, 1, 32, 32, dtype=torch.bool)
    masks[1, 0, 12:20, 8:18] = True
    with pytest.raises(FileNotFoundError):
        ed.run_edit_null_space_projection_xt(op="mid", block_idx=0, vis_num=2, mask_index=1, vis_num_pc=1, pca_rank=1)
    os.makedirs(os.path.join(ed.result_folder, "mask"))
    torch.save(masks, os.path.join(ed.result_folder, "mask", "mask.pt"))
    torch.manual_seed(5)
    x0 = ed.run_edit_null_space_projection_xt(op="mid", block_idx=0, vis_num=2, mask_index=1, vis_num_pc=1, pca_rank=1,
                                              null_space_projection=True, pca_rank_null=2)
    assert x0.dtype == torch.uint8 and tuple(x0.shape) == (5, 32, 32, 3)
    bdir = os.path.join(ed.result_folder, "basis", "local_basis-0.6T-pca-rank-1-select-mask1")
    for f in ("u-modify.pt", "vT-modify.pt", "u-null-null_space_rank_2.pt", "vT-null-null_space_rank_2.pt"):
        assert os.path.exists(os.path.join(bdir, f)), f
    torch.manual_seed(5)
    x0s = ed.run_edit_null_space_projection_xt_semantic(op="mid", block_idx=0, vis_num=2, mask_index=1, vis_num_pc=1, pca_rank=1,
                                                        null_space_projection=True, pca_rank_null=2, jacobian=True)
    assert tuple(x0s.shape) == (5, 32, 32, 3)
    sdir = os.path.join(ed.result_folder, "basis")
    pcs = [f for f in os.listdir(sdir) if f.startswith("Semantic_Edit_xt-") and f.endswith("-pc_000-vT.pt")]
    assert len(pcs) == 1
    v = torch.load(os.path.join(sdir, pcs[0]))
    assert tuple(v.shape) == (1, TINY_ADM.n) and abs(float(v.norm()) - 1.0) < 1e-4
    # reload through --vT_path: same frames
    ed2 = _edit(g, tmp_path, "bf16x3", vT_path=os.path.join(sdir, pcs[0]))
    os.makedirs(os.path.join(ed2.result_folder, "mask"), exist_ok=True)
    torch.manual_seed(5)
    x0r = ed2.run_edit_null_space_projection_xt_semantic(op="mid", block_idx=0, vis_num=2, mask_index=1, vis_num_pc=1, pca_rank=1)
    assert float((x0r.int() - x0s.int()).abs().float().mean()) < 1.0
    # sega ablation decodes the unedited x_t under three-way guidance
    ed3 = _edit(g, tmp_path, "bf16x3", ablation="sega")
    torch.manual_seed(5)
    xs = ed3.run_edit_null_space_projection_xt_semantic(op="mid", block_idx=0, vis_num=2, mask_index=1, vis_num_pc=1, pca_rank=1)
    assert tuple(xs.shape) == (1, 32, 32, 3)


def test_cli_shipped_if_script_on_the_standin(tmp_path, monkeypatch):
    """`python -m loco_edit_amd.main` with the argument list of scripts/main_T2I_DeepFloydIF_null_space_projection.sh
    (tests/golden/script_args.json) plus the deployment flags that replace what is out of scope (architecture preset,
    synthetic weights; SAM masks come from mask.pt): preset's DeepFloyd branch, the reference's result-folder layout,
    the Jacobian direction projected onto the null space, the saved --vT_path file."""
    import json
    from loco_edit_amd.main import main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = json.load(open(os.path.join(root, "tests", "golden", "script_args.json")))["main_T2I_DeepFloydIF_null_space_projection.sh"]
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("LOCO_PRECISION", "bf16x3")
    rdir = tmp_path / "runs" / "DeepFloyd-IF-Random-with_prompt" / "results" / "for_prompt_A photo of a man_cfg7.5_seed2628577915_standin"
    os.makedirs(rdir / "mask")
    masks = torch.zeros(13, 1, 32, 32, dtype=torch.bool)
    masks[12, 0, 12:20, 8:18] = True
    torch.save(masks, str(rdir / "mask" / "mask.pt"))
    x0 = main(argv + ["--device", DEV, "--unet_preset", "tiny_adm", "--synthetic_weights", "0"])
    assert x0.dtype == torch.uint8 and tuple(x0.shape) == (3, 32, 32, 3)      # vis_num 1: frames -S, 0, +S
    pcs = [f for f in os.listdir(rdir / "basis") if f.endswith("-pc_000-vT.pt")]
    assert len(pcs) == 1 and "edit_prompt-A photo of a man wearing glasses-select_mask12-null_space_projection_True_null_space_rank_5_null+(for-null)+(edit-null)" in pcs[0]
    v = torch.load(str(rdir / "basis" / pcs[0]))
    assert tuple(v.shape) == (1, TINY_ADM.n) and abs(float(v.norm()) - 1.0) < 1e-4
    assert any(f.endswith("_stage1.png") for f in os.listdir(rdir))


def test_pixel_space_tloco_with_text_cross_attention_vs_restatement(tmp_path):
    """The pixel-space class on a denoiser WITH text cross-attention stages (IF-I reads its T5 states through attention):
    prompt tokens go to `loco_set_context`, one context per CFG branch.  CFG noise (learned-variance half split off),
    x0_hat and a 3-iteration CFG-combined solve against the CPU restatement with `context=`."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import loco_oracle as orc
    import tloco_oracle as tl
    from loco_edit_amd.config import TINY_ADM_XATTN as cfg, synth_params
    from loco_edit_amd.tloco import EditDeepFloydIF
    os.environ.pop("WORLD_SIZE", None)
    g = torch.Generator().manual_seed(31)
    pe = {k: torch.randn(1, cfg.context_len, cfg.context_dim, generator=g) for k in ("for", "edit", "null")}
    args = Namespace(device=torch.device(DEV), dtype=torch.float32, seed=1, unet_config=cfg, synthetic_weights=0, ckpt_path="",
                     max_batch=8, precision="f32", dataset_name="Random", for_steps=100, use_yh_custom_scheduler=True,
                     guidance_scale=7.5, guidance_scale_edit=4.0, prompt_emb=pe, for_prompt="a", edit_prompt="b", edit_t=0.6,
                     sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)", ablation_method="null-space-proj",
                     mask_type="SAM", vT_path="", x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.5,
                     x_space_guidance_num_step=16, result_folder=str(tmp_path))
    ed = EditDeepFloydIF(args)
    assert ed.use_context
    ot = tl.OracleTLoco(orc.to_torch(synth_params(cfg, 0)), cfg, guidance_scale=7.5, guidance_scale_edit=4.0)
    x = torch.randn(1, 3, 32, 32, generator=g)
    t = ed.scheduler.timesteps[ed.edit_t_idx]
    F, E, N = pe["for"], pe["edit"], pe["null"]
    mask = torch.zeros(3, 32, 32, dtype=torch.bool); mask[:, 12:20, 8:18] = True
    with torch.no_grad():
        for mode in ("null+(for-null)+(edit-null)", "(for-edit)"):
            assert rel(ed._classifer_free_guidance(x.to(DEV), t, F, E, N, mode, True), ot.cfg_noise(x, t, F, E, N, mode)) < 1e-4
        assert rel(ed.get_x0(x.to(DEV), t, ed.edit_t_idx, F, E, N, mask=mask), ot.get_x0(x, t, F, E, N, mask=mask)) < 1e-4
    v0 = torch.randn(cfg.n, 2, generator=g)
    u, s, vT = ed.local_encoder_decoder_pullback_xt(x.to(DEV), t, ed.edit_t_idx, F, E, N, pca_rank=2, min_iter=3, max_iter=3,
                                                    mask=mask, mode="null+(for-null)", v0=v0.to(DEV), verbose=False)
    ou, os_, ovT = ot.pullback(x, t, F, E, N, 2, v0, min_iter=3, max_iter=3, mask=mask, mode="null+(for-null)")
    assert torch.allclose(s.cpu(), os_, rtol=1e-3) and cosrow(vT, ovT).min().item() > 0.999


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at ITS size: 64x64 pixels, four levels, attention at 32 / 16 / 8 (the 1024-token level included)
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_config5_64x64_vs_reference_fixture(prec, golden, tmp_path):
    """tests/golden/tloco_mid.pt: the reference's own `EditDeepFloydIF._classifer_free_guidance` / `get_x0` /
    `local_encoder_decoder_pullback_xt` (edit.py:1286-1373, 1566-1676) at 64x64 on config 5's geometry at a third of
    the width (MID_IF64: the size the CPU reference solves in minutes; oracle/make_golden_tloco.py --only mid)."""
    from loco_edit_amd.config import MID_IF64
    g = golden("tloco_mid")
    ed = _edit(g, tmp_path, prec, cfg=MID_IF64)
    tol = TOL[prec]
    x, t = g["x"].to(DEV), g["t"]
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    assert tuple(x.shape) == (1, 3, 64, 64) and ed.edit_t_idx == g["edit_t_idx"]
    xb = torch.cat([g["x"], g["x"].flip(-1)], dim=0).to(DEV)
    for mode, ref in g["eps_modes"].items():
        assert rel(ed._classifer_free_guidance(xb, t, F, E, N, mode, True), ref) < 4 * tol, mode
    assert rel(ed.get_x0(x, t, ed.edit_t_idx, F, E, N, mask=g["mask"]), g["x0_masked"]) < 4 * tol
    assert len(g["solver"]) == 2
    for mode, sv in g["solver"].items():
        u, s, vT = ed.local_encoder_decoder_pullback_xt(x, t, ed.edit_t_idx, F, E, N, pca_rank=3, min_iter=sv["n_iter"],
                                                        max_iter=sv["n_iter"], mask=sv["mask"], mode=mode,
                                                        v0=g["v0"].to(DEV), verbose=False)
        c = cosrow(vT, sv["vT"])
        print(f"[{prec}] config-5 geometry, mode {mode}: |cos| {c.tolist()}, s {s.tolist()}")
        assert ed.last_n_iter == sv["n_iter"] and torch.allclose(s.cpu(), sv["s"], rtol=1e-3), (mode, s.cpu(), sv["s"])
        assert c.min().item() > 0.999 and cosrow(u.T, sv["u"].T).min().item() > 0.999, mode


@pytest.mark.parametrize("preset", ["IF64_STANDIN", "IF64_XATTN_STANDIN", "SD64_STANDIN", "SD64_XATTN_STANDIN"])
def test_standin_presets_smoke_at_size(preset):
    """The round-2 / round-3 stand-in architectures stay selectable (`--unet_preset`): one at-size smoke test each -- the
    engine builds, the forward is finite, and J V / U^T J of the raw network are adjoint (two probes, default arithmetic).
    The full operator / solver checks at this width run on the real architectures (IF_I_M_UNET below, SD15_UNET in
    test_gpu_latent.py)."""
    import loco_edit_amd.config as C
    from loco_edit_amd.hip import LocoEngine
    cfg = getattr(C, preset)
    eng = LocoEngine(cfg, max_batch=2)
    eng.load_state_dict(C.synth_params(cfg, 0))
    eng.set_precision("bf16x3")
    gen = torch.Generator().manual_seed(5)
    if cfg.context_dim:
        eng.set_context(torch.randn(cfg.context_len, cfg.context_dim, generator=gen).to(DEV))
    x = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=gen).to(DEV)
    out = eng.unet_forward(x, 600.0)
    assert torch.isfinite(out).all() and float(out.abs().max()) > 0
    eng.pmp_primal(x, 600.0, 0.05, mask=None, use_et=True)
    V = torch.randn(2, cfg.n, generator=gen).to(DEV)
    JV = eng.pmp_jvp(V)
    U = torch.randn(2, JV.shape[1], generator=gen).to(DEV)
    JtU = eng.pmp_vjp(U)
    lhs, rhs = (JV.double() * U.double()).sum(dim=1), (V.double() * JtU.double()).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1)).double()).max().item() < 2e-4


@pytest.mark.parametrize("preset", ["IF_I_M_UNET"])   # (the stand-ins ran here until the IF architecture itself did: 21 - 27 s each; smoke tests above)
def test_config5_full_width_operator_and_solver_at_size(preset, tmp_path):
    """Config 5 at its stated size AND width (64x64, 192 x (1,2,3,4), 3 ResBlocks per level, attention at 32 / 16 / 8 with
    64-channel heads, learned variance).  `IF_I_M_UNET` is the architecture the shipped script names (DeepFloyd/IF-I-M-v1.0:
    GELU, (skip + h) / sqrt 2, attention over [text ; image] keys, the 77 x 4096 T5 states conditioned on the host; synthetic
    weights); the two stand-ins are the guided-diffusion U-Net with the text through the time embedding only / through
    T5-shaped 77 x 4096 cross-attention stages: no CPU
    reference finishes at this width, so the checks are the size-independent ones -- adjointness <J V, U> = <V, J^T U>,
    linearity, J V against a central finite difference of get_x0, and a 12-iteration top-5 solve whose rows are
    orthonormal, whose s descends and equals ||J v_i|| computed by an independent product."""
    import loco_edit_amd.config as C
    from loco_edit_amd.tloco import EditDeepFloydIF
    cfg = getattr(C, preset)
    os.environ.pop("WORLD_SIZE", None)
    gen = torch.Generator().manual_seed(31)
    if cfg.context_dim:
        width = cfg.encoder_dim if cfg.encoder_dim > 0 else cfg.context_dim     # the IF U-Net takes the text encoder's own states
        pe = {k: torch.randn(1, cfg.context_len, width, generator=gen) for k in ("for", "edit", "null")}
    else:
        pe = {k: torch.randn(1, 7, 16, generator=gen) for k in ("for", "edit", "null")}
    args = Namespace(device=torch.device(DEV), dtype=torch.float32, seed=1, unet_config=cfg, synthetic_weights=0, ckpt_path="",
                     max_batch=8, precision="f32", dataset_name="Random", for_steps=100, use_yh_custom_scheduler=True,
                     guidance_scale=7.5, guidance_scale_edit=4.0, prompt_emb=pe, for_prompt="a", edit_prompt="b", edit_t=0.6,
                     sampling_mode=False, tilda_v_score_type="null+(for-null)+(edit-null)", ablation_method="null-space-proj",
                     mask_type="SAM", vT_path="", x_space_guidance_edit_step=1.0, x_space_guidance_scale=0.5,
                     x_space_guidance_num_step=16, result_folder=str(tmp_path))
    ed = EditDeepFloydIF(args)      # exact-fp32 arithmetic for the operator checks: the finite difference needs the digits
    F, E, N = pe["for"], pe["edit"], pe["null"]
    x = torch.randn(1, 3, 64, 64, generator=gen).to(DEV)
    t = ed.scheduler.timesteps[ed.edit_t_idx]
    mask = torch.zeros(3, 64, 64, dtype=torch.bool); mask[:, 24:40, 16:36] = True
    mode = "null+(for-null)+(edit-null)"
    ed._bind_all(F, E, N)
    op = ed._operator(x, t, mask, mode)
    V = torch.randn(3, cfg.n, generator=gen).to(DEV)
    U = (torch.randn(3, cfg.n, generator=gen) * mask.reshape(1, -1)).to(DEV)
    JV, JtU = op.jvp(V), op.vjp(U)
    lhs, rhs = (JV * U).sum(dim=1), (V * JtU).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1))).max().item() < 2e-4
    assert float(JV[:, ~mask.reshape(-1).to(DEV)].abs().max()) == 0.0
    comb = (2.0 * V[0] - 0.5 * V[1] + 0.25 * V[2])[None].contiguous()
    assert rel(op.jvp(comb)[0], 2.0 * JV[0] - 0.5 * JV[1] + 0.25 * JV[2]) < 2e-4
    jv_n = op.gather(op.jvp((V[0] / V[0].norm())[None].contiguous()))
    v = (V[0] / V[0].norm()).view(1, 3, 64, 64)
    errs = []
    for h in (2e-2, 1e-2):                       # truncation error ~h^2, fp32 round-off of x0_hat (|x0_hat| ~ 50 under guidance 7.5) ~1/h
        fd = (ed.get_x0(x + h * v, t, ed.edit_t_idx, F, E, N, mask=mask, mode=mode)
              - ed.get_x0(x - h * v, t, ed.edit_t_idx, F, E, N, mask=mask, mode=mode)) / (2 * h)
        errs.append(rel(jv_n, fd))
    print(f"[{preset}] J v vs central differences: rel err {errs} at h = 2e-2, 1e-2")
    assert min(errs) < 2.5e-2
    for eng in ed.branches.values():             # the solve in the default arithmetic
        eng.set_precision("bf16x3")
    v0 = torch.randn(cfg.n, 5, generator=gen).to(DEV)
    u, s, vT = ed.local_encoder_decoder_pullback_xt(x, t, ed.edit_t_idx, F, E, N, pca_rank=5, min_iter=12, max_iter=12,
                                                    mask=mask, mode=mode, v0=v0, verbose=False)
    assert ed.last_n_iter == 12 and tuple(vT.shape) == (5, cfg.n) and tuple(u.shape) == (int(mask.sum()), 5)
    vd = vT.double()
    assert (vd @ vd.T - torch.eye(5, device=DEV, dtype=torch.float64)).abs().max().item() < 2e-6
    assert bool((s[:-1] >= s[1:] * (1 - 1e-5)).all()) and bool(torch.isfinite(vT).all())
    op = ed._operator(x, t, mask, mode)
    assert torch.allclose(op.jvp(vT.contiguous()).norm(dim=1).cpu(), s.cpu(), rtol=2e-2)


def test_cfg_branches_side_by_side_equal_the_serial_order(golden, tmp_path):
    """The CFG branches on their own HIP streams (tloco.BranchStreams, the default) against one branch after the other:
    guided noise on a batch, J V and J^T U of the three-branch operator and a 4-iteration solve -- bit-identical (the branches
    are separate engine contexts; only the order of independent launches changes)."""
    g = golden("tloco_tiny")
    ed = _edit(g, tmp_path, "bf16x3")
    assert ed.branch_streams.enabled and len(ed.branch_streams.side) == 2
    x, t = g["x"].to(DEV), g["t"]
    F, E, N = g["for_e"], g["edit_e"], g["null_e"]
    mode = "null+(for-null)+(edit-null)"
    gen = torch.Generator().manual_seed(9)
    mask = g["mask"].to(DEV)
    V = torch.randn(4, ed.engine.n, generator=gen).to(DEV)
    U = torch.randn(4, ed.engine.n, generator=gen).to(DEV)
    xb = torch.cat([x, 0.5 * x.flip(-1), x + 0.1])

    def everything():
        e = ed._classifer_free_guidance(xb, t, F, E, N, mode, True)
        op = ed._operator(x, t, mask, mode)
        jv, jtu = op.jvp(V), op.vjp(U)
        u, s, vT = ed.local_encoder_decoder_pullback_xt(x, t, ed.edit_t_idx, F, E, N, pca_rank=3, min_iter=4, max_iter=4, mask=mask,
                                                        mode=mode, v0=g["v0"].to(DEV), verbose=False)
        torch.cuda.synchronize()
        return [z.clone() for z in (e, jv, jtu, u, s, vT)]
    side = everything()
    ed.branch_streams.enabled = False
    serial = everything()
    for a, b in zip(side, serial):
        assert torch.equal(a, b)
