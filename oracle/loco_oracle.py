"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (CPU, fp32) restatement of the LOCO-Edit null-space-projection
hot path of the reference (ChicyChen/LOCO-Edit @ 2024-10-22).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``loco-edit_amd/`` never does.

Parity status: PINNED for the unconditional DDPM path (BASELINE.json configs
0-1) -- ``oracle/make_golden.py`` imports the reference itself in the build
container (stub modules for the absent torchvision/diffusers/skimage), checks
every function below against the reference's own output and commits the
input/output vectors under ``tests/golden/`` (the reference has no tests or
golden vectors of its own).

Every function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# Denoiser A: Ho-DDPM U-Net (reference src/models/ddpm/diffusion.py)
# --------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """[sin, cos] sinusoid with divisor (half-1) -- diffusion.py:783-804."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def _swish(x):  # diffusion.py:806-808
    return x * torch.sigmoid(x)


def _gn(p, name, x, cfg):  # diffusion.py:810-811 (GroupNorm 32, eps 1e-6)
    return F.group_norm(x, cfg.gn_groups, p[name + ".weight"], p[name + ".bias"], cfg.gn_eps)


def _resblock(p, name, x, temb, cfg):
    """diffusion.py:855-912."""
    h = _gn(p, name + ".norm1", x, cfg)
    h = _swish(h)
    h = F.conv2d(h, p[name + ".conv1.weight"], p[name + ".conv1.bias"], padding=1)
    h = h + F.linear(_swish(temb), p[name + ".temb_proj.weight"], p[name + ".temb_proj.bias"])[:, :, None, None]
    h = _gn(p, name + ".norm2", h, cfg)
    h = _swish(h)
    h = F.conv2d(h, p[name + ".conv2.weight"], p[name + ".conv2.bias"], padding=1)
    if (name + ".nin_shortcut.weight") in p:
        x = F.conv2d(x, p[name + ".nin_shortcut.weight"], p[name + ".nin_shortcut.bias"])
    return x + h


def _attn(p, name, x, cfg):
    """Single-head self-attention over H*W tokens -- diffusion.py:914-966."""
    h = _gn(p, name + ".norm", x, cfg)
    q = F.conv2d(h, p[name + ".q.weight"], p[name + ".q.bias"])
    k = F.conv2d(h, p[name + ".k.weight"], p[name + ".k.bias"])
    v = F.conv2d(h, p[name + ".v.weight"], p[name + ".v.bias"])
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    h = F.conv2d(h, p[name + ".proj_out.weight"], p[name + ".proj_out.bias"])
    return x + h


def unet_forward(p: Dict[str, torch.Tensor], cfg, x: torch.Tensor, t: torch.Tensor,
                 trace: Optional[dict] = None) -> torch.Tensor:
    """eps_theta(x, t) -- PullBackDDPM.forward, diffusion.py:145-200.
    ``trace`` (tests only) collects each block's output under its module name."""
    def rec(name, v):
        if trace is not None:
            trace[name] = v.detach().clone()
        return v
    t = t.reshape(1) if t.dim() == 0 else t
    temb = timestep_embedding(t.to(torch.float32), cfg.ch)
    temb = F.linear(temb, p["temb.dense.0.weight"], p["temb.dense.0.bias"])
    temb = _swish(temb)
    temb = F.linear(temb, p["temb.dense.1.weight"], p["temb.dense.1.bias"])
    nres = len(cfg.ch_mult)
    res = cfg.resolution
    hs = [rec("conv_in", F.conv2d(x, p["conv_in.weight"], p["conv_in.bias"], padding=1))]
    for lvl in range(nres):
        for b in range(cfg.num_res_blocks):
            h = rec(f"down.{lvl}.block.{b}", _resblock(p, f"down.{lvl}.block.{b}", hs[-1], temb, cfg))
            if res in cfg.attn_resolutions:
                h = rec(f"down.{lvl}.attn.{b}", _attn(p, f"down.{lvl}.attn.{b}", h, cfg))
            hs.append(h)
        if lvl != nres - 1:
            # Downsample: pad (0,1,0,1) then 3x3 stride 2 -- diffusion.py:834-853
            h = F.pad(hs[-1], (0, 1, 0, 1))
            hs.append(rec(f"down.{lvl}.downsample.conv",
                          F.conv2d(h, p[f"down.{lvl}.downsample.conv.weight"],
                                   p[f"down.{lvl}.downsample.conv.bias"], stride=2)))
            res //= 2
    h = hs[-1]
    h = rec("mid.block_1", _resblock(p, "mid.block_1", h, temb, cfg))
    h = rec("mid.attn_1", _attn(p, "mid.attn_1", h, cfg))
    h = rec("mid.block_2", _resblock(p, "mid.block_2", h, temb, cfg))
    for lvl in reversed(range(nres)):
        for b in range(cfg.num_res_blocks + 1):
            h = rec(f"up.{lvl}.block.{b}",
                    _resblock(p, f"up.{lvl}.block.{b}", torch.cat([h, hs.pop()], dim=1), temb, cfg))
            if res in cfg.attn_resolutions:
                h = rec(f"up.{lvl}.attn.{b}", _attn(p, f"up.{lvl}.attn.{b}", h, cfg))
        if lvl != 0:
            # Upsample: nearest x2 then 3x3 -- diffusion.py:816-832
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = rec(f"up.{lvl}.upsample.conv",
                    F.conv2d(h, p[f"up.{lvl}.upsample.conv.weight"], p[f"up.{lvl}.upsample.conv.bias"], padding=1))
            res *= 2
    h = _swish(_gn(p, "norm_out", h, cfg))
    return F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)


# --------------------------------------------------------------------------
# Latent decoder: the network behind ``self.vae.decode(z).sample`` in the reference's Stable Diffusion path
# (edit.py:750, 770-771).  diffusers' AutoencoderKL is un-vendored; its decoder is the latent-diffusion ``Decoder``,
# i.e. the up half of the module tree above without skip inputs and with ``temb_channels = 0`` (ResnetBlock without
# ``temb_proj``), restated here from that published structure.  Parity of the architecture is unpinned (no diffusers,
# no weights offline); the engine is checked against this restatement.
# --------------------------------------------------------------------------
def _resblock_noemb(p, name, x, cfg):
    h = F.conv2d(_swish(_gn(p, name + ".norm1", x, cfg)), p[name + ".conv1.weight"], p[name + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(p, name + ".norm2", h, cfg)), p[name + ".conv2.weight"], p[name + ".conv2.bias"], padding=1)
    if (name + ".nin_shortcut.weight") in p:
        x = F.conv2d(x, p[name + ".nin_shortcut.weight"], p[name + ".nin_shortcut.bias"])
    return x + h


def decoder_forward(p: Dict[str, torch.Tensor], cfg, z: torch.Tensor, trace: Optional[dict] = None) -> torch.Tensor:
    """z [B, z_ch, R, R] -> image [B, out_ch, R * 2^(levels-1), same]."""
    def rec(name, v):
        if trace is not None:
            trace[name] = v.detach().clone()
        return v
    nlev = len(cfg.ch_mult)
    res = cfg.resolution
    h = rec("post_quant_conv", F.conv2d(z, p["post_quant_conv.weight"], p["post_quant_conv.bias"]))   # AutoencoderKL.decode
    h = rec("conv_in", F.conv2d(h, p["conv_in.weight"], p["conv_in.bias"], padding=1))
    h = rec("mid.block_1", _resblock_noemb(p, "mid.block_1", h, cfg))
    h = rec("mid.attn_1", _attn(p, "mid.attn_1", h, cfg))
    h = rec("mid.block_2", _resblock_noemb(p, "mid.block_2", h, cfg))
    for lvl in reversed(range(nlev)):
        for b in range(cfg.num_res_blocks + 1):
            h = rec(f"up.{lvl}.block.{b}", _resblock_noemb(p, f"up.{lvl}.block.{b}", h, cfg))
            if res in cfg.attn_resolutions:
                h = rec(f"up.{lvl}.attn.{b}", _attn(p, f"up.{lvl}.attn.{b}", h, cfg))
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = rec(f"up.{lvl}.upsample.conv",
                    F.conv2d(h, p[f"up.{lvl}.upsample.conv.weight"], p[f"up.{lvl}.upsample.conv.bias"], padding=1))
            res *= 2
    h = _swish(_gn(p, "norm_out", h, cfg))
    return F.conv2d(h, p["conv_out.weight"], p["conv_out.bias"], padding=1)


def encoder_forward(p: Dict[str, torch.Tensor], cfg, x: torch.Tensor) -> torch.Tensor:
    """image [B, 3, R, R] -> moments [B, 2 z_ch, R / 2^(levels-1), same] = quant_conv(Encoder(x)): the latent-diffusion
    `Encoder` + `quant_conv` behind `vae.encode(x0).latent_dist` of the reference's latent inversion (edit.py:594-597;
    un-vendored diffusers AutoencoderKL, restated from the published latent-diffusion module): conv_in; per level
    num_res_blocks ResnetBlocks and, except on the last, Downsample = pad (0,1,0,1) + conv3 stride 2; mid block / attention /
    block; GroupNorm, swish, conv_out."""
    nlev = len(cfg.ch_mult)
    h = F.conv2d(x, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    for lvl in range(nlev):
        for b in range(cfg.num_res_blocks):
            h = _resblock_noemb(p, f"down.{lvl}.block.{b}", h, cfg)
        if lvl != nlev - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), p[f"down.{lvl}.downsample.conv.weight"], p[f"down.{lvl}.downsample.conv.bias"],
                         stride=2)
    h = _resblock_noemb(p, "mid.block_1", h, cfg)
    h = _attn(p, "mid.attn_1", h, cfg)
    h = _resblock_noemb(p, "mid.block_2", h, cfg)
    h = F.conv2d(_swish(_gn(p, "norm_out", h, cfg)), p["conv_out.weight"], p["conv_out.bias"], padding=1)
    return F.conv2d(h, p["quant_conv.weight"], p["quant_conv.bias"])


# --------------------------------------------------------------------------
# Denoiser B: guided-diffusion / P2 U-Net (reference src/models/guided_diffusion/unet.py, P2_DICT)
# --------------------------------------------------------------------------
def timestep_embedding_adm(t: torch.Tensor, dim: int) -> torch.Tensor:
    """[cos, sin] sinusoid with divisor half -- guided_diffusion/nn.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _adm_gn(p, name, x, cfg):  # GroupNorm32(32, C), eps 1e-5, computed in fp32 -- nn.py:17-19
    return F.group_norm(x.float(), cfg.gn_groups, p[name + ".weight"], p[name + ".bias"], cfg.gn_eps).type(x.dtype)


def _act(cfg):
    """SiLU (guided-diffusion, latent-diffusion) or the exact erf GELU of the DeepFloyd-IF U-Net (`act_fn = "gelu"`)."""
    return F.gelu if getattr(cfg, "act", "silu") == "gelu" else F.silu


def _adm_resblock(p, name, x, emb, cfg, up=False, down=False):
    """ResBlock with use_scale_shift_norm=True -- unet.py:145-258.  DeepFloyd-IF variant (cfg.act / cfg.res_scale;
    deepfloyd_if UNetModel = diffusers ResnetBlock2D with `time_embedding_norm="scale_shift"`, `skip_time_act`,
    `output_scale_factor = sqrt 2`, un-vendored): GELU for SiLU, the embedding arrives activated once for all blocks
    (the same act(emb) this block computes), output (skip + h) / sqrt 2."""
    act = _act(cfg)
    h = act(_adm_gn(p, name + ".in_layers.0", x, cfg))
    if up:      # Upsample(channels, False): nearest x2 on both branches -- :195-197, 239-244
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:  # Downsample(channels, False): avg_pool2d(2) on both branches -- :198-200
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, p[name + ".in_layers.2.weight"], p[name + ".in_layers.2.bias"], padding=1)
    emb_out = F.linear(act(emb), p[name + ".emb_layers.1.weight"], p[name + ".emb_layers.1.bias"])[:, :, None, None]
    if getattr(cfg, "scale_shift_norm", True):
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _adm_gn(p, name + ".out_layers.0", h, cfg) * (1 + scale) + shift     # :250-254
    else:                                                                        # :255-257
        h = _adm_gn(p, name + ".out_layers.0", h + emb_out, cfg)
    h = F.conv2d(act(h), p[name + ".out_layers.3.weight"], p[name + ".out_layers.3.bias"], padding=1)
    if (name + ".skip_connection.weight") in p:
        x = F.conv2d(x, p[name + ".skip_connection.weight"], p[name + ".skip_connection.bias"])
    return (x + h) * getattr(cfg, "res_scale", 1.0)


def _adm_attn(p, name, x, cfg):
    """AttentionBlock + QKVAttentionLegacy -- unet.py:261-307, 330-356."""
    b, c, hh, ww = x.shape
    xr = x.reshape(b, c, -1)
    qkv = F.conv1d(_adm_gn(p, name + ".norm", xr, cfg), p[name + ".qkv.weight"], p[name + ".qkv.bias"])
    n_heads = c // cfg.num_head_channels
    ch = c
    ch = (3 * c) // (3 * n_heads)
    q, k, v = qkv.reshape(b * n_heads, ch * 3, hh * ww).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, hh * ww)
    h = F.conv1d(a, p[name + ".proj_out.weight"], p[name + ".proj_out.bias"])
    return (xr + h).reshape(b, c, hh, ww)


def _if_attn(p, name, x, context, cfg):
    """AttentionBlock of the DeepFloyd-IF U-Net (deepfloyd_if/model/unet.py AttentionBlock + QKVAttention, the GLIDE
    text2im attention; diffusers `Attention(added_kv_proj_dim=..., cross_attention_norm="group_norm")` with
    `AttnAddedKVProcessor`, un-vendored): the text states [B, L, D] pass the block's GroupNorm (over [D][L]) and a Conv1d to
    2C channels, split per head into [k_h | v_h] and CONCATENATED IN FRONT of the image keys / values -- one softmax over
    L + T columns; q, k, v of the image tokens as in QKVAttentionLegacy; out = x + proj_out(a)."""
    b, c, hh, ww = x.shape
    xr = x.reshape(b, c, -1)
    qkv = F.conv1d(_adm_gn(p, name + ".norm", xr, cfg), p[name + ".qkv.weight"], p[name + ".qkv.bias"])
    nh = c // cfg.num_head_channels
    ch = c // nh
    q, k, v = qkv.reshape(b * nh, ch * 3, hh * ww).split(ch, dim=1)
    e = context.transpose(1, 2)                                                         # [B, D, L]
    e = F.group_norm(e.float(), cfg.gn_groups, p[name + ".norm_encoder.weight"], p[name + ".norm_encoder.bias"], cfg.gn_eps)
    e = F.conv1d(e, p[name + ".encoder_kv.weight"], p[name + ".encoder_kv.bias"])         # [B, 2C, L]
    ek, ev = e.reshape(b * nh, ch * 2, -1).split(ch, dim=1)
    k = torch.cat([ek, k], dim=-1)
    v = torch.cat([ev, v], dim=-1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, hh * ww)
    h = F.conv1d(a, p[name + ".proj_out.weight"], p[name + ".proj_out.bias"])
    return (xr + h).reshape(b, c, hh, ww)


def if_text_conditioning(p, cfg, states):
    """Host-side conditioning of the IF U-Net from the text encoder's states [B, L, E] (diffusers UNet2DConditionModel.forward
    with `encoder_hid_dim_type="text_proj"`, `addition_embed_type="text"`; deepfloyd_if `encoder_proj` / `encoder_pooling`):
    context = Linear(states) [B, L, D] for the attention blocks, aug = LayerNorm(Linear(AttentionPooling(LayerNorm(states))))
    [B, 4 ch] added to the time embedding.  AttentionPooling: a class token mean(states) + positional_embedding queries
    [token ; states] with `num_heads = E / 64` heads (IF: 64 heads of 64)."""
    states = states.float()
    ctx = F.linear(states, p["encoder_proj.weight"], p["encoder_proj.bias"])
    E = states.shape[-1]
    x = F.layer_norm(states, (E,), p["encoder_pooling.0.weight"], p["encoder_pooling.0.bias"], 1e-5)
    b = x.shape[0]
    dph = min(64, E // 2)
    nh = E // dph
    cls = x.mean(dim=1, keepdim=True) + p["encoder_pooling.1.positional_embedding"]
    xs = torch.cat([cls, x], dim=1)                                                     # [B, L + 1, E]

    def shape(z):    # [B, N, E] -> [B * heads, d, N]
        return z.reshape(b, -1, nh, dph).transpose(1, 2).reshape(b * nh, -1, dph).transpose(1, 2)
    q = shape(F.linear(cls, p["encoder_pooling.1.q_proj.weight"], p["encoder_pooling.1.q_proj.bias"]))
    k = shape(F.linear(xs, p["encoder_pooling.1.k_proj.weight"], p["encoder_pooling.1.k_proj.bias"]))
    v = shape(F.linear(xs, p["encoder_pooling.1.v_proj.weight"], p["encoder_pooling.1.v_proj.bias"]))
    sc = 1 / math.sqrt(math.sqrt(dph))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * sc, k * sc).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, 1).transpose(1, 2)[:, 0, :]     # [B, E]
    a = F.linear(a, p["encoder_pooling.2.weight"], p["encoder_pooling.2.bias"])
    aug = F.layer_norm(a, (a.shape[-1],), p["encoder_pooling.3.weight"], p["encoder_pooling.3.bias"], 1e-5)
    return ctx, aug


def _ldm_spatial_transformer(p, name, x, context, cfg):
    """SpatialTransformer of latent-diffusion / Stable Diffusion v1 (ldm/modules/attention.py: SpatialTransformer,
    BasicTransformerBlock, CrossAttention, FeedForward / GEGLU; un-vendored by the reference, which reaches it through
    diffusers' UNet2DConditionModel -- the same arithmetic under other parameter names), depth 1:
    GroupNorm(32, eps 1e-6) -> 1x1 proj_in -> tokens [B, HW, C] -> x + attn1(LN(x)); x + attn2(LN(x), context);
    x + Linear(GEGLU(LN(x))) -> 1x1 proj_out -> + input.  Heads = cfg.num_heads (or C / cfg.num_head_channels), scale = head_dim^-1/2, to_q/k/v without
    bias, LayerNorm eps 1e-5, GEGLU: proj to 8C, value * gelu(gate) (erf form)."""
    b, c, hh, ww = x.shape
    # v1: a fixed head count per block (num_heads = 8); 2.x: a fixed head width (num_head_channels = 64)
    nh = cfg.num_heads if cfg.num_heads > 0 else c // cfg.num_head_channels
    d = c // nh
    h = F.group_norm(x, cfg.gn_groups, p[name + ".norm.weight"], p[name + ".norm.bias"], 1e-6)
    h = F.conv2d(h, p[name + ".proj_in.weight"], p[name + ".proj_in.bias"])
    h = h.reshape(b, c, hh * ww).transpose(1, 2)                                       # [B, T, C]
    tb = name + ".transformer_blocks.0"

    def ln(n, z):
        return F.layer_norm(z, (c,), p[f"{tb}.{n}.weight"], p[f"{tb}.{n}.bias"], 1e-5)

    def attention(an, z, ctx):
        q = F.linear(z, p[f"{tb}.{an}.to_q.weight"])
        k = F.linear(ctx, p[f"{tb}.{an}.to_k.weight"])
        v = F.linear(ctx, p[f"{tb}.{an}.to_v.weight"])
        q, k, v = (u.reshape(b, -1, nh, d).permute(0, 2, 1, 3) for u in (q, k, v))     # [B, H, N, d]
        w = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * (d ** -0.5), dim=-1)
        o = torch.einsum("bhij,bhjd->bhid", w, v).permute(0, 2, 1, 3).reshape(b, -1, c)
        return F.linear(o, p[f"{tb}.{an}.to_out.0.weight"], p[f"{tb}.{an}.to_out.0.bias"])

    z = ln("norm1", h)
    h = h + attention("attn1", z, z)
    h = h + attention("attn2", ln("norm2", h), context)
    f = F.linear(ln("norm3", h), p[f"{tb}.ff.net.0.proj.weight"], p[f"{tb}.ff.net.0.proj.bias"])
    val, gate = f.chunk(2, dim=-1)
    h = h + F.linear(val * F.gelu(gate), p[f"{tb}.ff.net.2.weight"], p[f"{tb}.ff.net.2.bias"])
    h = h.transpose(1, 2).reshape(b, c, hh, ww)
    return x + F.conv2d(h, p[name + ".proj_out.weight"], p[name + ".proj_out.bias"])


def _adm_xattn(p, name, x, context, cfg):
    """Text cross-attention stage of the stand-in denoisers (no counterpart in guided_diffusion; the reference's
    cross-attention lives in un-vendored diffusers blocks): h = x + proj(softmax(q^T k / sqrt(d)) v) with q from
    GN(x) per image token and k, v Linear projections of the [B, L, D] encoder states, heads of num_head_channels."""
    b, c, hh, ww = x.shape
    nh = c // cfg.num_head_channels if cfg.num_head_channels > 0 else 1
    ch = c // nh
    h = _adm_gn(p, name + ".norm", x, cfg).reshape(b, c, -1)
    q = F.conv1d(h, p[name + ".q.weight"], p[name + ".q.bias"])                        # [B, C, T]
    k = F.linear(context, p[name + ".k.weight"], p[name + ".k.bias"]).transpose(1, 2)   # [B, C, L]
    v = F.linear(context, p[name + ".v.weight"], p[name + ".v.bias"]).transpose(1, 2)
    q, k, v = (z.reshape(b * nh, ch, -1) for z in (q, k, v))
    w_ = torch.softmax(torch.einsum("bct,bcl->btl", q, k) * (ch ** -0.5), dim=-1)
    o = torch.einsum("btl,bcl->bct", w_, v).reshape(b, c, -1)
    o = F.conv1d(o, p[name + ".proj_out.weight"], p[name + ".proj_out.bias"])
    return x + o.reshape(b, c, hh, ww)


def unet_forward_adm(p, cfg, x, t, trace: Optional[dict] = None, emb_add: Optional[torch.Tensor] = None,
                     full: bool = False, context: Optional[torch.Tensor] = None):
    """UNetModel.forward returning the eps half -- unet.py:636-684, constructor :398-617.
    ``emb_add`` [B, 4*ch] (or [4*ch]) is added to the time embedding where the class embedding goes
    (``emb = emb + self.label_emb(y)``, unet.py:660-662); ``full`` keeps the learned-variance channels."""
    def rec(name, v):
        if trace is not None:
            trace[name] = v.detach().clone()
        return v

    def attn(name, h):     # self-attention [+ the text cross-attention stage of the stand-in denoisers]
        if getattr(cfg, "transformer_depth", 0) > 0:
            ctx = context if context.dim() == 3 else context[None]
            return _ldm_spatial_transformer(p, name, h, ctx.expand(h.shape[0], -1, -1), cfg)
        if getattr(cfg, "added_kv", False):
            ctx = context if context.dim() == 3 else context[None]
            return _if_attn(p, name, h, ctx.expand(h.shape[0], -1, -1), cfg)
        h = _adm_attn(p, name, h, cfg)
        if getattr(cfg, "context_dim", 0) > 0:
            ctx = context if context.dim() == 3 else context[None]
            h = _adm_xattn(p, name + ".xattn", h, ctx.expand(h.shape[0], -1, -1), cfg)
        return h
    t = t.reshape(1) if t.dim() == 0 else t
    emb = timestep_embedding_adm(t.to(torch.float32), cfg.ch)
    emb = F.linear(emb, p["time_embed.0.weight"], p["time_embed.0.bias"])
    emb = F.linear(_act(cfg)(emb), p["time_embed.2.weight"], p["time_embed.2.bias"])
    if emb_add is not None:
        emb = emb + emb_add
    hs = []
    h = rec("input_blocks.0.0", F.conv2d(x, p["input_blocks.0.0.weight"], p["input_blocks.0.0.bias"], padding=1))
    hs.append(h)
    res_px = cfg.resolution
    ib = 1
    nlev = len(cfg.ch_mult)
    for lvl in range(nlev):
        for _ in range(cfg.num_res_blocks):
            h = rec(f"input_blocks.{ib}.0", _adm_resblock(p, f"input_blocks.{ib}.0", h, emb, cfg))
            if res_px in cfg.attn_resolutions:
                h = rec(f"input_blocks.{ib}.1", attn(f"input_blocks.{ib}.1", h))
            hs.append(h); ib += 1
        if lvl != nlev - 1:
            if getattr(cfg, "resblock_updown", True):
                h = rec(f"input_blocks.{ib}.0", _adm_resblock(p, f"input_blocks.{ib}.0", h, emb, cfg, down=True))
            else:           # Downsample(use_conv=True): conv3 stride 2 padding 1 -- unet.py:113-142
                h = rec(f"input_blocks.{ib}.0", F.conv2d(h, p[f"input_blocks.{ib}.0.op.weight"], p[f"input_blocks.{ib}.0.op.bias"],
                                                         stride=2, padding=1))
            hs.append(h); ib += 1
            res_px //= 2
    h = rec("middle_block.0", _adm_resblock(p, "middle_block.0", h, emb, cfg))
    h = rec("middle_block.1", attn("middle_block.1", h))
    h = rec("middle_block.2", _adm_resblock(p, "middle_block.2", h, emb, cfg))
    ob = 0
    for lvl in reversed(range(nlev)):
        for i in range(cfg.num_res_blocks + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            h = rec(f"output_blocks.{ob}.0", _adm_resblock(p, f"output_blocks.{ob}.0", h, emb, cfg))
            j = 1
            if res_px in cfg.attn_resolutions:
                h = rec(f"output_blocks.{ob}.{j}", attn(f"output_blocks.{ob}.{j}", h)); j += 1
            if lvl and i == cfg.num_res_blocks:
                if getattr(cfg, "resblock_updown", True):
                    h = rec(f"output_blocks.{ob}.{j}", _adm_resblock(p, f"output_blocks.{ob}.{j}", h, emb, cfg, up=True))
                else:       # Upsample(use_conv=True): nearest x2, conv3 -- unet.py:83-110
                    h = F.interpolate(h, scale_factor=2, mode="nearest")
                    h = rec(f"output_blocks.{ob}.{j}", F.conv2d(h, p[f"output_blocks.{ob}.{j}.conv.weight"],
                                                                p[f"output_blocks.{ob}.{j}.conv.bias"], padding=1))
                res_px *= 2
            ob += 1
    h = _act(cfg)(_adm_gn(p, "out.0", h, cfg))
    h = F.conv2d(h, p["out.2.weight"], p["out.2.bias"], padding=1)
    if cfg.learn_sigma and not full:
        h = torch.split(h, h.shape[1] // 2, dim=1)[0]      # et only (unet.py:680-684)
    return h


def denoiser(p, cfg, x, t, trace=None):
    return unet_forward_adm(p, cfg, x, t, trace) if getattr(cfg, "arch", "ddpm") == "adm" else unet_forward(p, cfg, x, t, trace)


# --------------------------------------------------------------------------
# Scheduler (reference src/utils/utils.py:305-461)
# --------------------------------------------------------------------------
class Scheduler:
    """YHCustomScheduler restated -- utils.py:305-423 (linear schedule only,
    which is what ``preset`` forces for the unconditional models,
    define_argparser.py:233)."""

    t_max = 999

    def __init__(self):
        betas = torch.linspace(0.0001, 0.02, 1000, dtype=torch.float64)  # utils.py:385-391,408-409
        self.betas = betas.to(torch.float32)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(torch.float32)  # :401-403
        self.timesteps = None
        self.timesteps_next = None

    def set_timesteps(self, n: int, is_inversion: bool = False):  # utils.py:316-329
        seq = torch.linspace(0, 1, n) * self.t_max
        if is_inversion:
            seq = seq + 1e-6
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]])
            self.timesteps = seq_prev[1:]
            self.timesteps_next = seq[1:]
        else:
            seq_prev = torch.cat([torch.tensor([-1.0]), seq[:-1]])
            self.timesteps = torch.flip(seq[1:], dims=[0])
            self.timesteps_next = torch.flip(seq_prev[1:], dims=[0])

    def get_timesteps(self, t):  # utils.py:331-337
        return self.timesteps_next[torch.where(self.timesteps == t)]

    def alpha_at(self, t) -> torch.Tensor:
        """``extract``: alpha-bar gathered at floor(t) -- utils.py:444-461."""
        return self.alphas_cumprod[int(torch.as_tensor(t).long().item())]

    def step(self, et, t, xt, eta: float = 0.0, noise: Optional[torch.Tensor] = None):
        """utils.py:342-383 (learn_sigma False).  ``noise`` injects the randn_like draw."""
        idx = self.timesteps.tolist().index(float(t))
        t_next = self.timesteps_next[idx]
        at = self.alpha_at(t)
        at_next = self.alpha_at(t_next)
        p_xt = (xt - et * (1 - at).sqrt()) / at.sqrt()
        if eta == 0:
            xt_next = at_next.sqrt() * p_xt + (1 - at_next).sqrt() * et
        else:
            sigma = ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            if noise is None:
                noise = torch.randn_like(xt)
            xt_next = at_next.sqrt() * p_xt + (1 - at_next - eta * sigma ** 2).sqrt() * et + eta * sigma * noise
        return xt_next, p_xt


# --------------------------------------------------------------------------
# Edit pipeline (reference src/modules/edit.py:2034-2625)
# --------------------------------------------------------------------------
class OracleEdit:
    def __init__(self, params: Dict[str, torch.Tensor], cfg, for_steps=100, inv_steps=100,
                 edit_t=0.6, performance_boosting_t=0.2):
        self.p = params
        self.cfg = cfg
        self.sched = Scheduler()
        self.for_steps, self.inv_steps = for_steps, inv_steps
        self.sched.set_timesteps(for_steps)
        ts = self.sched.timesteps
        self.edit_t_idx = int((ts - edit_t * 1000).abs().argmin())  # edit.py:2071-2072
        self.performance_boosting_t_idx = (
            int((ts - performance_boosting_t * 1000).abs().argmin())
            if performance_boosting_t > 0 else 1000)  # edit.py:2073

    def unet(self, x, t):
        return denoiser(self.p, self.cfg, x, t)

    # -- edit.py:2369-2391
    def get_x0(self, t, x, mask=None):
        et = self.unet(x, t)
        at = self.sched.alpha_at(t)
        p_xt = (x - et * (1 - at).sqrt()) / at.sqrt()
        if mask is not None:
            p_xt = p_xt[:, mask]
        return p_xt

    # -- edit.py:2394-2403
    def get_et(self, t, x, mask=None):
        et = self.unet(x, t)
        if mask is not None:
            et = et[:, mask]
        return et

    # -- edit.py:2117-2167 (98 of 99 steps, eta = 0)
    @torch.no_grad()
    def ddim_inversion(self, x0):
        self.sched.set_timesteps(self.inv_steps, is_inversion=True)
        ts = self.sched.timesteps
        xt = x0
        for i, t in enumerate(ts):
            if i == len(ts) - 1:
                break
            xt, _ = self.sched.step(self.unet(xt, t), t, xt, eta=0)
        return xt

    # -- edit.py:2508-2614 (CPU bounce buffer / chunking are no-ops numerically)
    @torch.no_grad()
    def ddim_forwardsteps(self, xt, t_start_idx, t_end_idx, performance_boosting=False,
                          noises: Optional[Sequence[torch.Tensor]] = None):
        self.sched.set_timesteps(self.for_steps)
        ts = self.sched.timesteps
        for i, t in enumerate(ts):
            if t_end_idx == i:
                return xt, t, i
            elif i < t_start_idx:
                continue
            eta = 1 if (performance_boosting and self.performance_boosting_t_idx <= i
                        and self.performance_boosting_t_idx != len(ts) - 1) else 0
            nz = None if (noises is None or eta == 0) else noises[i]
            xt, _ = self.sched.step(self.unet(xt, t), t, xt, eta=eta, noise=nz)
        return xt

    # -- edit.py:2406-2504 block power iteration on J^T J
    def pullback(self, x, t, pca_rank, v0: torch.Tensor, min_iter=10, max_iter=100,
                 convergence_threshold=1e-3, mask=None, noise=False, chunk_size=25,
                 verbose=False):
        """``v0`` is the [n, k] Gaussian matrix the reference draws at edit.py:2435
        (injected so results are reproducible); it is QR-orthonormalised exactly as
        at :2436.  Returns (u [L,k], s [k], vT [k,n], n_iter)."""
        c, hh, ww = x.shape[1:]
        n = c * hh * ww
        num_chunk = pca_rank // chunk_size if pca_rank % chunk_size == 0 else pca_rank // chunk_size + 1
        a = torch.tensor(0.0)
        q, _ = torch.linalg.qr(v0.to(torch.float32))
        v = q.T.reshape(-1, c, hh, ww)
        fn = self.get_et if noise else self.get_x0
        n_done = 0
        for i in range(max_iter):
            v_prev = v.detach().clone()
            u = []
            for vi in v.chunk(num_chunk):
                g = lambda a_: fn(t, x + a_ * vi, mask=mask)
                u.append(torch.func.jacfwd(g, argnums=0, randomness="error")(a).detach())  # u_i = J v_i
            u = torch.cat(u, dim=0)
            if mask is None:
                g2 = lambda x_: torch.einsum("bcwh,icwh->b", u, fn(t, x_, mask=mask))
            else:
                g2 = lambda x_: torch.einsum("bl,il->b", u, fn(t, x_, mask=mask))
            v_ = torch.autograd.functional.jacobian(g2, x).reshape(-1, n)  # rows u_i^T J
            _, s, v = torch.linalg.svd(v_, full_matrices=False)
            v = v.reshape(-1, c, hh, ww)
            n_done = i + 1
            if verbose:
                print(f"power method : {i}-th step convergence : ", torch.dist(v_prev, v).item())
            if torch.allclose(v_prev, v, atol=convergence_threshold) and (i > min_iter):
                break
        u = u.reshape(u.shape[0], -1).T.detach()
        return u, s.sqrt().detach(), v.reshape(-1, n).detach(), n_done

    # -- edit.py:2313-2323
    @staticmethod
    def project(vT_modify, vT_null=None, pca_rank_null=None):
        if vT_null is None:
            return vT_modify / vT_modify.norm(dim=1, keepdim=True)
        vT_null = vT_null[:pca_rank_null, :]
        vT = (vT_null.T @ (vT_null @ vT_modify.T)).T
        vT = vT_modify - vT
        return vT / vT.norm(dim=1, keepdim=True)

    # -- edit.py:2339-2363 and :2618-2625
    @staticmethod
    def edit_batch(xt, vk_row, scale, num_step, vis_num, edit_step=1.0):
        xts = {}
        for direction in (1, -1):
            vk = direction * vk_row.view(-1, *xt.shape[1:])
            lst = [xt.clone()]
            for _ in range(num_step):
                lst.append(lst[-1] + scale * edit_step * vk)
            b = torch.cat(lst, dim=0)
            b = b[[0, -1], :] if vis_num == 1 else b[::(b.size(0) // vis_num)]
            xts[direction] = b
        return torch.cat([xts[-1].flip(dims=[0])[:-1], xts[1]], dim=0)


# --------------------------------------------------------------------------
# Explicit J / J^T products (used by tests as an independent check of the HIP
# tangent / cotangent passes, and by bench.py's cpu_baseline leg)
# --------------------------------------------------------------------------
def jvp_x0(ed: OracleEdit, x, t, V, mask=None, noise=False):
    """U = J V with J = d x0_hat[mask] / d x (forward mode, as edit.py:2451-2455)."""
    fn = ed.get_et if noise else ed.get_x0
    a = torch.tensor(0.0)
    g = lambda a_: fn(t, x + a_ * V, mask=mask)
    return torch.func.jacfwd(g)(a).detach()


def vjp_x0(ed: OracleEdit, x, t, U, mask=None, noise=False):
    """A = U^T J (reverse mode, as edit.py:2460-2480); U is [k, L] or [k,c,h,w]."""
    fn = ed.get_et if noise else ed.get_x0
    if mask is None:
        g2 = lambda x_: torch.einsum("bcwh,icwh->b", U, fn(t, x_, mask=mask))
    else:
        g2 = lambda x_: torch.einsum("bl,il->b", U, fn(t, x_, mask=mask))
    return torch.autograd.functional.jacobian(g2, x).reshape(U.shape[0], -1).detach()


def to_torch(params_np) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v.copy()) for k, v in params_np.items()}
