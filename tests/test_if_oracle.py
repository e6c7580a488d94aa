"""CPU: the DeepFloyd-IF stage-I architecture (config.IF_I_M_UNET) -- presets and routing, the size of the network, the
host-side text conditioning against the restatement, and the diffusers naming of its checkpoint.  The restatement
(oracle/loco_oracle.py `_if_attn`, `if_text_conditioning`, `_adm_resblock` with cfg.act / cfg.res_scale) is written from the
published module trees; neither diffusers nor deepfloyd_if is installed and there are no weights: parity unpinned."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loco_edit_amd  # noqa: E402,F401
import loco_oracle as orc  # noqa: E402
from loco_edit_amd import checkpoints as K, config as C, define_argparser  # noqa: E402


def test_if_presets_routing_and_size(tmp_path, monkeypatch):
    cfg = C.IF_I_M_UNET
    assert (cfg.act, cfg.added_kv, cfg.encoder_dim, cfg.context_dim, cfg.context_len) == ("gelu", True, 4096, 768, 77)
    assert abs(cfg.res_scale - 1 / math.sqrt(2)) < 1e-12
    shapes = C.param_shapes(cfg)
    host = sum(math.prod(s) for k, s in shapes.items() if k.startswith(("encoder_proj.", "encoder_pooling.")))
    total = sum(math.prod(s) for s in shapes.values())
    # IF-I-M is published as a 400 M-parameter model; the same tree at 704 channels gives IF-I-XL's 4.3 B
    assert (total - host, host, total) == (314_955_078, 56_650_752, 371_605_830)
    xl, lg = C.if_stage1_config("XL"), C.if_stage1_config("L")
    assert (xl.ch, xl.context_dim, lg.ch) == (704, 2816, 320) and C.if_stage1_config("M") is cfg
    assert 4.2e9 < sum(math.prod(s) for s in C.param_shapes(xl).values()) < 4.4e9       # published: 4.3 B
    assert 0.85e9 < sum(math.prod(s) for s in C.param_shapes(lg).values()) < 1.0e9      # published: 0.9 B
    import json
    monkeypatch.chdir(tmp_path)
    argv = json.load(open(os.path.join(ROOT, "tests", "golden", "script_args.json")))["main_T2I_DeepFloydIF_null_space_projection.sh"]
    a = define_argparser.preset(define_argparser.parse_args(argv + ["--device", "cpu"]))
    assert a.is_DeepFloyd_IF_diffusion and a.unet_config is C.IF_I_M_UNET and a.model_name.split("-")[2] == "M"
    b = define_argparser.preset(define_argparser.parse_args(argv + ["--device", "cpu", "--unet_preset", "if64_standin"]))
    assert b.unet_config is C.IF64_STANDIN
    i = argv.index("--model_name")
    argv_l = argv[:i + 1] + ["DeepFloyd/IF-I-L-v1.0"] + argv[i + 2:]
    assert define_argparser.preset(define_argparser.parse_args(argv_l + ["--device", "cpu"])).unet_config.ch == 320
    import pytest
    # IF-I-XL: one shared parameter store for the three guidance branches (loco_fork, the default) -> accepted; three independently
    # loaded 4.3 B-parameter contexts (LOCO_CFG_FORK=0) are refused up front
    argv_xl = argv[:i + 1] + ["DeepFloyd/IF-I-XL-v1.0"] + argv[i + 2:]
    monkeypatch.delenv("LOCO_CFG_FORK", raising=False)
    assert define_argparser.preset(define_argparser.parse_args(argv_xl + ["--device", "cpu"])).unet_config.ch == 704
    monkeypatch.setenv("LOCO_CFG_FORK", "0")
    with pytest.raises(SystemExit, match="IF-I-XL"):
        define_argparser.preset(define_argparser.parse_args(argv_xl + ["--device", "cpu"]))
    monkeypatch.delenv("LOCO_CFG_FORK")
    with pytest.raises(ValueError):
        K.hf_if_unet_to_native({}, C.FFHQ_P2)


def test_if_text_conditioner_equals_restatement():
    from loco_edit_amd.tloco import IFTextConditioner
    for cfg in (C.TINY_IF, C.MID_IF):
        params = C.synth_params(cfg, 3)
        g = torch.Generator().manual_seed(1)
        states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
        context, aug = IFTextConditioner(params, cfg, "cpu")(states)
        ctx_ref, aug_ref = orc.if_text_conditioning(orc.to_torch(params), cfg, states)
        assert tuple(context.shape) == (cfg.context_len, cfg.context_dim) and tuple(aug.shape) == (4 * cfg.ch,)
        assert torch.allclose(context, ctx_ref[0], atol=1e-5) and torch.allclose(aug, aug_ref[0], atol=1e-5)


def _diffusers_added_kv_attention(a, x, states, heads, groups, eps):
    """diffusers `AttnAddedKVProcessor` on an `Attention(added_kv_proj_dim=D, cross_attention_norm="group_norm", bias=True)`
    restated on its own parameter names: tokens [B, T, C]; heads are contiguous channel blocks; key = cat([add_k, to_k])."""
    b, c, hh, ww = x.shape
    h = x.reshape(b, c, -1)
    h = F.group_norm(h, groups, a["group_norm.weight"], a["group_norm.bias"], eps).transpose(1, 2)
    e = F.group_norm(states.transpose(1, 2), groups, a["norm_cross.weight"], a["norm_cross.bias"], eps).transpose(1, 2)

    def split(z):     # [B, N, C] -> [B, heads, N, d]
        return z.reshape(b, -1, heads, c // heads).permute(0, 2, 1, 3)
    q = split(F.linear(h, a["to_q.weight"], a["to_q.bias"]))
    k = torch.cat([split(F.linear(e, a["add_k_proj.weight"], a["add_k_proj.bias"])), split(F.linear(h, a["to_k.weight"], a["to_k.bias"]))], dim=2)
    v = torch.cat([split(F.linear(e, a["add_v_proj.weight"], a["add_v_proj.bias"])), split(F.linear(h, a["to_v.weight"], a["to_v.bias"]))], dim=2)
    w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c // heads), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(b, -1, c)
    o = F.linear(o, a["to_out.0.weight"], a["to_out.0.bias"]).transpose(1, 2).reshape(b, c, hh, ww)
    return o + x


def test_if_checkpoint_in_diffusers_naming_loads_and_means_the_same():
    cfg = C.TINY_IF
    sd = {k: torch.as_tensor(v) for k, v in C.synth_params(cfg, 5).items()}
    hf = K.native_to_hf_if_unet(sd, cfg)
    assert K.is_hf_if_unet(hf) and not K.is_hf_if_unet(sd)
    assert "down_blocks.0.downsamplers.0.conv1.weight" in hf and "up_blocks.0.upsamplers.0.norm2.bias" in hf
    assert "add_embedding.pool.positional_embedding" in hf and "encoder_hid_proj.weight" in hf
    back = K.normalize_unet_state_dict(hf, cfg)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    # the per-head interleave of q / k / v and of the text keys / values: the diffusers processor on the diffusers names
    # equals the native block on the native names
    name, hname = "middle_block.1", "mid_block.attentions.0"
    c = cfg.ch * cfg.ch_mult[-1]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, c, 8, 8, generator=g)
    states = torch.randn(2, cfg.context_len, cfg.context_dim, generator=g)
    a = {k[len(hname) + 1:]: v for k, v in hf.items() if k.startswith(hname + ".")}
    ref = _diffusers_added_kv_attention(a, x, states, c // cfg.num_head_channels, cfg.gn_groups, cfg.gn_eps)
    got = orc._if_attn(sd, name, x, states, cfg)
    assert torch.allclose(got, ref, atol=2e-5), float((got - ref).abs().max())


def test_if_restatement_uses_gelu_scale_and_joint_softmax():
    """Each switch changes the restated network; the text states enter through a GroupNorm and share the image keys' softmax."""
    cfg = C.TINY_IF
    p = orc.to_torch(C.synth_params(cfg, 2))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, 32, 32, generator=g)
    states = torch.randn(1, cfg.context_len, cfg.encoder_dim, generator=g)
    ctx, aug = orc.if_text_conditioning(p, cfg, states)
    t = torch.tensor(300.0)
    base = orc.unet_forward_adm(p, cfg, x, t, emb_add=aug, context=ctx)
    for change in ({"act": "silu"}, {"res_scale": 1.0}):
        other = C.UNetConfig(**{**cfg.__dict__, **change})
        assert (orc.unet_forward_adm(p, other, x, t, emb_add=aug, context=ctx) - base).abs().max() > 1e-3
    # the states pass a GroupNorm inside every block (norm_encoder): a rescaled context changes nothing, another one does
    assert (orc.unet_forward_adm(p, cfg, x, t, emb_add=aug, context=2.0 * ctx) - base).abs().max() < 1e-4
    other_ctx = ctx + 0.5 * torch.randn(ctx.shape, generator=g)
    assert (orc.unet_forward_adm(p, cfg, x, t, emb_add=aug, context=other_ctx) - base).abs().max() > 1e-3
    # one softmax over [text ; image]: not the sum of a text attention and an image attention
    h = torch.randn(1, 64, 8, 8, generator=g)
    joint = orc._if_attn(p, "middle_block.1", h, ctx, cfg) - h
    no_text = {**p, "middle_block.1.encoder_kv.weight": torch.zeros_like(p["middle_block.1.encoder_kv.weight"]),
               "middle_block.1.encoder_kv.bias": torch.full_like(p["middle_block.1.encoder_kv.bias"], -1e4)}
    # (keys at -1e4 in every channel take no probability for queries with a positive channel sum and all of it otherwise:
    # the two results differ, and the text columns demonstrably take part in the normalisation)
    assert (orc._if_attn(no_text, "middle_block.1", h, ctx, cfg) - h - joint).abs().max() > 1e-3
