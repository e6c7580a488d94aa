"""GPU tests of the DeepFloyd-IF stage-I denoiser (config 5's architecture, `config.IF_I_M_UNET` and its small instances):
exact GELU in the norm -> activation -> conv chains and the time embedding, ResBlock outputs (skip + h) / sqrt 2, attention
over [text ; image] keys in one softmax (`added_kv`), the text conditioning computed on the host.  The engine against the
torch restatement in oracle/loco_oracle.py (`_if_attn`, `if_text_conditioning`, `_adm_resblock` with cfg.act /
cfg.res_scale): forward, J V (torch.func.jvp) and U^T J (autograd), in the three conv arithmetics.  The restatement itself
is unpinned against diffusers / deepfloyd_if (neither is installed, no weights): bars as in test_gpu_parity.py.  The
full-width network (IF_I_M_UNET) runs through the T-LOCO class in test_gpu_tloco.py::test_config5_full_width_operator_and_solver_at_size."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loco_edit_amd  # noqa: E402,F401
import loco_oracle as orc  # noqa: E402
from loco_edit_amd.config import MID_IF, TINY_IF, synth_params  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"f32": 2e-5, "bf16x3": 2e-4, "f16": 2e-2}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _engine(cfg, params, states, max_batch):
    from loco_edit_amd.hip import LocoEngine
    from loco_edit_amd.tloco import IFTextConditioner
    eng = LocoEngine(cfg, max_batch=max_batch, device=torch.device(DEV))
    eng.load_state_dict(params)                       # skips the host-side encoder_proj / encoder_pooling entries
    context, aug = IFTextConditioner(params, cfg, DEV)(states)
    eng.set_context(context)
    eng.set_cond(aug)
    return eng, context, aug


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "f16"])
def test_if_denoiser_forward_jvp_vjp_vs_restatement(prec):
    cfg = TINY_IF
    params = synth_params(cfg, 2)
    p = orc.to_torch(params)
    g = torch.Generator().manual_seed(11)
    states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    eng, context, aug = _engine(cfg, params, states, 4)
    ctx_ref, aug_ref = orc.if_text_conditioning(p, cfg, states)
    assert rel(context, ctx_ref[0]) < 1e-5 and rel(aug, aug_ref[0]) < 1e-5          # host conditioning vs the restatement
    eng.set_precision(prec)
    x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=g)
    t = 417.0
    f = lambda x_: orc.unet_forward_adm(p, cfg, x_, torch.tensor(t), emb_add=aug_ref, context=ctx_ref)
    with torch.no_grad():
        ref = f(x)
    assert rel(eng.unet_forward(x.to(DEV), t), ref) < TOL[prec]
    xb = torch.cat([x, 0.5 * x, x + 0.1], dim=0)                                     # a batch through one launch list
    with torch.no_grad():
        refb = orc.unet_forward_adm(p, cfg, xb, torch.full((3,), t), emb_add=aug_ref, context=ctx_ref)
    assert rel(eng.unet_forward(xb.to(DEV), t), refb) < TOL[prec]
    V = torch.randn(3, eng.n, generator=g)
    JV = torch.stack([torch.func.jvp(f, (x,), (v.view_as(x),))[1].reshape(-1) for v in V])
    eng.pmp_primal(x.to(DEV), t, 0.5, None, use_et=True)
    U = eng.pmp_jvp(V.to(DEV))
    assert rel(U, JV) < TOL[prec] * 5
    Uc = torch.randn(3, eng.n_out, generator=g)
    A = eng.pmp_vjp(Uc.to(DEV))
    xx = x.clone().requires_grad_(True)
    out = f(xx).reshape(-1)
    Aref = torch.stack([torch.autograd.grad((out * u).sum(), xx, retain_graph=True)[0].reshape(-1) for u in Uc])
    assert rel(A, Aref) < TOL[prec] * 5
    lhs, rhs = (U.double().cpu() * Uc.double()).sum(), (V.double() * A.double().cpu()).sum()
    assert abs(lhs - rhs) / abs(lhs) < (1e-4 if prec != "f16" else 2e-2)
    # the x0 combination and a mask, as the solver uses the operator
    mask = torch.zeros(3, cfg.resolution, cfg.resolution, dtype=torch.bool); mask[:, 8:24, 4:20] = True
    at = 0.37
    eng.pmp_primal(x.to(DEV), t, at, mask.to(DEV))
    m = mask.reshape(1, -1).float()
    ref0 = m * (V / at ** 0.5 - (1 - at) ** 0.5 / at ** 0.5 * JV)
    assert rel(eng.pmp_jvp(V.to(DEV)), ref0) < TOL[prec] * 5


def test_if_denoiser_text_changes_the_output_and_needs_the_context():
    """The prompt reaches the network through BOTH routes (attention keys / values and the pooled embedding); a context-free
    engine refuses to run."""
    from loco_edit_amd.hip import LocoEngine
    cfg = TINY_IF
    params = synth_params(cfg, 2)
    g = torch.Generator().manual_seed(4)
    s0 = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    s1 = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    x = torch.randn(1, 3, 32, 32, generator=g).to(DEV)
    eng, c0, a0 = _engine(cfg, params, s0, 2)
    e00 = eng.unet_forward(x, 100.0).clone()
    from loco_edit_amd.tloco import IFTextConditioner
    c1, a1 = IFTextConditioner(params, cfg, DEV)(s1)
    eng.set_context(c1)
    e10 = eng.unet_forward(x, 100.0).clone()
    eng.set_cond(a1)
    e11 = eng.unet_forward(x, 100.0).clone()
    assert rel(e10, e00) > 1e-3 and rel(e11, e10) > 1e-3
    bare = LocoEngine(cfg, max_batch=1, device=torch.device(DEV))
    bare.load_state_dict(params)
    with pytest.raises(RuntimeError):
        bare.unet_forward(x, 100.0)


def test_if_denoiser_mid_size_1024_token_level():
    """Config 5's geometry at a third of the width (`MID_IF`: 64 x 64, four levels, three attention levels incl. 1024 image
    tokens + 77 text states -> 1152 score columns, the streamed softmax rows): forward vs the restatement, J V / J^T U
    adjointness and J V vs autodiff for one probe."""
    cfg = MID_IF
    params = synth_params(cfg, 1)
    p = orc.to_torch(params)
    g = torch.Generator().manual_seed(12)
    states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    eng, _, _ = _engine(cfg, params, states, 3)
    ctx_ref, aug_ref = orc.if_text_conditioning(p, cfg, states)
    x = torch.randn(1, 3, 64, 64, generator=g)
    t = 594.0
    f = lambda x_: orc.unet_forward_adm(p, cfg, x_, torch.tensor(t), emb_add=aug_ref, context=ctx_ref)
    with torch.no_grad():
        ref = f(x)
    for prec in ("bf16x3", "f32"):
        eng.set_precision(prec)
        assert rel(eng.unet_forward(x.to(DEV), t), ref) < TOL[prec]
    eng.set_precision("bf16x3")
    eng.pmp_primal(x.to(DEV), t, 0.5, None, use_et=True)
    V = torch.randn(3, eng.n, generator=g)
    U = torch.randn(3, eng.n, generator=g).to(DEV)
    JV, JtU = eng.pmp_jvp(V.to(DEV)), eng.pmp_vjp(U)
    jv_ref = torch.func.jvp(f, (x,), (V[0].view_as(x),))[1].reshape(1, -1)
    assert rel(JV[0:1], jv_ref) < TOL["bf16x3"] * 5
    lhs, rhs = (JV.double() * U.double()).sum(dim=1), (V.to(DEV).double() * JtU.double()).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1)).double()).max().item() < 2e-4


def test_cli_shipped_if_script_on_the_if_architecture(tmp_path, monkeypatch):
    """`python -m loco_edit_amd.main` with the argument list of scripts/main_T2I_DeepFloydIF_null_space_projection.sh
    (tests/golden/script_args.json) on a small instance of the IF architecture (`--unet_preset tiny_if`, synthetic weights,
    seeded text states; SAM masks from mask.pt): the DeepFloyd branch of preset, the reference's result folder
    (`..._seed<seed>_M`, edit.py:1204-1206), the projected direction saved under the reference's file name."""
    import json
    from loco_edit_amd.main import main
    argv = json.load(open(os.path.join(ROOT, "tests", "golden", "script_args.json")))["main_T2I_DeepFloydIF_null_space_projection.sh"]
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setenv("LOCO_PRECISION", "bf16x3")
    rdir = tmp_path / "runs" / "DeepFloyd-IF-Random-with_prompt" / "results" / "for_prompt_A photo of a man_cfg7.5_seed2628577915_M"
    os.makedirs(rdir / "mask")
    masks = torch.zeros(13, 1, 32, 32, dtype=torch.bool)
    masks[12, 0, 12:20, 8:18] = True
    torch.save(masks, str(rdir / "mask" / "mask.pt"))
    x0 = main(argv + ["--device", DEV, "--unet_preset", "tiny_if", "--synthetic_weights", "0"])
    assert x0.dtype == torch.uint8 and tuple(x0.shape) == (3, 32, 32, 3)
    # `--dtype fp16` (how the reference loads its IF pipeline) selects the f16 conv arithmetic over fp32 tensors
    monkeypatch.delenv("LOCO_PRECISION", raising=False)
    from loco_edit_amd import define_argparser
    from loco_edit_amd.tloco import EditDeepFloydIF
    a16 = define_argparser.preset(define_argparser.parse_args(argv + ["--device", DEV, "--unet_preset", "tiny_if", "--synthetic_weights", "0",
                                                                      "--dtype", "fp16"]))
    ed = EditDeepFloydIF(a16)
    assert ed.engine.get_precision() == "f16" and ed.dtype == torch.float32
    pcs = [f for f in os.listdir(rdir / "basis") if f.endswith("-pc_000-vT.pt")]
    assert len(pcs) == 1
    v = torch.load(str(rdir / "basis" / pcs[0]))
    assert tuple(v.shape) == (1, TINY_IF.n) and abs(float(v.norm()) - 1.0) < 1e-4


def test_if_attention_flash_kernels_equal_the_strided_products(monkeypatch):
    """The tangent / cotangent of the [text ; image] attention on the flash kernels (attn_flash.hip TXT: two text key blocks
    ahead of the image blocks, no [T x (128 + T)] matrix per probe) against the strided products over one score matrix
    (LOCO_FLASH_ATTN=0), same arithmetic class: 1024- and 256-token levels on the flash kernels, the 64-token level on the
    products either way."""
    from loco_edit_amd.hip import LocoEngine
    from loco_edit_amd.tloco import IFTextConditioner
    cfg = MID_IF
    params = synth_params(cfg, 1)
    g = torch.Generator().manual_seed(21)
    states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    context, aug = IFTextConditioner(params, cfg, DEV)(states)
    x = torch.randn(1, 3, 64, 64, generator=g).to(DEV)
    V = torch.randn(4, cfg.n, generator=g).to(DEV)
    U = torch.randn(4, cfg.n, generator=g).to(DEV)
    res = {}
    for flash in ("1", "0"):
        monkeypatch.setenv("LOCO_FLASH_ATTN", flash)
        eng = LocoEngine(cfg, max_batch=4, device=torch.device(DEV))
        eng.load_state_dict(params)
        eng.set_context(context); eng.set_cond(aug)
        eng.set_precision("bf16x3")
        eng.pmp_primal(x, 300.0, 0.5, None, use_et=True)
        res[flash] = (eng.pmp_jvp(V).clone(), eng.pmp_vjp(U).clone())
        del eng
    assert rel(res["1"][0], res["0"][0]) < 1e-4 and rel(res["1"][1], res["0"][1]) < 1e-4
    assert not torch.equal(res["1"][0], res["0"][0])          # two different code paths did run


def test_if_i_m_denoiser_at_size_vs_restatement():
    """`config.IF_I_M_UNET` at full width (315 M U-Net parameters, synthetic weights) with the weights arriving the way a
    diffusers checkpoint stores them (UNet2DConditionModel names, separate to_q / to_k / to_v / add_k_proj / add_v_proj):
    forward against the CPU restatement in both arithmetics, J V / J^T U adjointness and linearity of the raw network."""
    from loco_edit_amd.checkpoints import native_to_hf_if_unet, normalize_unet_state_dict
    from loco_edit_amd.config import IF_I_M_UNET
    from loco_edit_amd.hip import LocoEngine
    from loco_edit_amd.tloco import IFTextConditioner
    cfg = IF_I_M_UNET
    params = synth_params(cfg, 0)
    p = orc.to_torch(params)
    g = torch.Generator().manual_seed(8)
    states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    x = torch.randn(1, 3, 64, 64, generator=g)
    t = 742.0
    ctx_ref, aug_ref = orc.if_text_conditioning(p, cfg, states)
    with torch.no_grad():
        ref = orc.unet_forward_adm(p, cfg, x, torch.tensor(t), emb_add=aug_ref, context=ctx_ref)
    sd = normalize_unet_state_dict(native_to_hf_if_unet(p, cfg), cfg)
    del p
    eng = LocoEngine(cfg, max_batch=3, device=torch.device(DEV))
    eng.load_state_dict(sd)
    context, aug = IFTextConditioner(sd, cfg, DEV)(states)
    del sd
    assert rel(context, ctx_ref[0]) < 1e-5 and rel(aug, aug_ref[0]) < 1e-5
    eng.set_context(context); eng.set_cond(aug)
    for prec in ("bf16x3", "f32"):
        eng.set_precision(prec)
        e = rel(eng.unet_forward(x.to(DEV), t), ref)
        print(f"IF-I-M U-Net forward at size, {prec} vs CPU restatement: rel err {e:.2e}")
        assert e < TOL[prec]
    eng.set_precision("bf16x3")
    eng.pmp_primal(x.to(DEV), t, 0.5, None, use_et=True)
    V = torch.randn(3, cfg.n, generator=g).to(DEV)
    U = torch.randn(3, cfg.n, generator=g).to(DEV)
    JV, JtU = eng.pmp_jvp(V), eng.pmp_vjp(U)
    lhs, rhs = (JV.double() * U.double()).sum(dim=1), (V.double() * JtU.double()).sum(dim=1)
    assert ((lhs - rhs).abs() / (JV.norm(dim=1) * U.norm(dim=1)).double()).max().item() < 2e-4
    comb = eng.pmp_jvp((V[0:1] * 0.5 - V[1:2] * 2.0).contiguous())
    assert rel(comb, JV[0:1] * 0.5 - JV[1:2] * 2.0) < 1e-3
