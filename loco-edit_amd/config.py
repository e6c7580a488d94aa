"""Denoiser configuration, checkpoint key layout and the deterministic weight
synthesiser.

The key layout is the ``state_dict`` of the reference's vendored Ho-DDPM U-Net
(reference ``src/models/ddpm/diffusion.py:22-126``: ``temb.dense.N``,
``conv_in``, ``down.L.block.B.{norm1,conv1,temb_proj,norm2,conv2,nin_shortcut}``,
``down.L.attn.B.{norm,q,k,v,proj_out}``, ``down.L.downsample.conv``,
``mid.{block_1,attn_1,block_2}``, ``up.L...``, ``norm_out``, ``conv_out``), so a
checkpoint saved from that module loads here unchanged.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np


@dataclass(frozen=True)
class UNetConfig:
    """Mirrors ``config.model`` / ``config.data`` of ``configs/custom_celeba_ddpm.yml``."""
    resolution: int = 256
    in_channels: int = 3
    out_ch: int = 3
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4, 4)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = (16,)
    gn_groups: int = 32
    gn_eps: float = 1e-6
    # "ddpm": Ho et al. U-Net (models/ddpm/diffusion.py); "adm": guided-diffusion / P2 U-Net
    # (models/guided_diffusion/unet.py with P2_DICT: scale-shift norm, resblock up/down, legacy multi-head attention)
    # "dec": latent decoder (conv_in, mid, up levels, norm_out/conv_out; no skips, no time embedding) -- the
    # `vae.decode` network of the reference's Stable Diffusion path (edit.py:750, 770-771); `resolution` is then the
    # LATENT resolution and the output is [out_ch, resolution * 2^(levels-1), same]
    arch: str = "ddpm"
    num_head_channels: int = -1     # adm: channels per attention head (P2: 64)
    learn_sigma: bool = False       # adm: network emits 2*out_ch channels, eps = the first out_ch (unet.py:680-684)
    # adm, text-to-image stand-ins: context_dim > 0 puts a text cross-attention stage behind every attention block
    # (queries from the image tokens, keys / values from the context_len x context_dim encoder states of the prompt)
    context_dim: int = 0
    context_len: int = 0
    # adm, the latent-diffusion (Stable Diffusion v1) U-Net = the same guided-diffusion skeleton with
    # `use_scale_shift_norm=False` (unet.py:255-257: h + emb_out), `resblock_updown=False` (Downsample / Upsample with a
    # 3x3 conv, unet.py:83-142), `num_heads` fixed per block instead of a head width, and a SpatialTransformer
    # (GroupNorm -> 1x1 proj_in -> [LayerNorm -> self-attention, LayerNorm -> cross-attention over the prompt states,
    # LayerNorm -> GEGLU feed-forward, each with its residual] -> 1x1 proj_out -> + input) where ADM has its AttentionBlock
    scale_shift_norm: bool = True
    resblock_updown: bool = True
    num_heads: int = -1
    transformer_depth: int = 0
    # adm, the DeepFloyd-IF stage-I denoiser (deepfloyd_if UNetModel = diffusers UNet2DConditionModel with
    # ResnetDownsampleBlock2D / SimpleCrossAttn*Block2D): exact GELU instead of SiLU ("efficient activation": the blocks'
    # embedding projections read gelu(emb) once), ResBlock output (skip + h) * res_scale, and attention blocks whose keys /
    # values are [text ; image] in one softmax (`added_kv`: the context_len x context_dim states, already through the
    # model's encoder_proj, pass the block's norm_encoder GroupNorm and its encoder_kv projection).  `encoder_dim` is the
    # width of the text encoder's states (T5-XXL: 4096) that the host-side conditioning reads (tloco.IFTextConditioner:
    # encoder_proj -> context, encoder_pooling -> the embedding added to the time embedding)
    act: str = "silu"
    res_scale: float = 1.0
    added_kv: bool = False
    encoder_dim: int = 0

    @property
    def temb_ch(self) -> int:
        return self.ch * 4

    @property
    def n(self) -> int:
        return self.in_channels * self.resolution * self.resolution

    @property
    def out_resolution(self) -> int:
        if self.arch == "dec":
            return self.resolution << (len(self.ch_mult) - 1)
        if self.arch == "enc":          # image -> moments of the latent posterior, `resolution` = the IMAGE resolution
            return self.resolution >> (len(self.ch_mult) - 1)
        return self.resolution

    @property
    def n_out(self) -> int:
        return self.out_ch * self.out_resolution * self.out_resolution


# configs[0..1] of BASELINE.json: google/ddpm-celebahq-256 architecture
CELEBA_DDPM = UNetConfig()
# parity-test sizes the CPU oracle finishes in seconds
TINY_DDPM = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=2,
                       attn_resolutions=(16,))
MID_DDPM = UNetConfig(resolution=64, ch=32, ch_mult=(1, 1, 2), num_res_blocks=1,
                      attn_resolutions=(16,))
# config 2 of BASELINE.json: FFHQ-P2 (script_util.py:166-190 P2_DICT + create_model :379-435)
FFHQ_P2 = UNetConfig(resolution=256, ch=128, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=1, attn_resolutions=(16,),
                     gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=True)
# BASELINE config 5 stand-in: pixel-space 64x64 conditional denoiser in the shape class of DeepFloyd IF-I (4 levels, 3
# ResBlocks per level, attention at 32/16/8, 64-channel heads, learned variance); the real IF U-Net is diffusers'
# UNet2DConditionModel with T5 cross-attention (un-vendored) -- here the text enters through the time embedding only
IF64_STANDIN = UNetConfig(resolution=64, ch=192, ch_mult=(1, 2, 3, 4), num_res_blocks=3, attn_resolutions=(32, 16, 8),
                          gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=True)
# BASELINE config 4 stand-ins (Stable Diffusion v1.5 shape class; the real networks are diffusers' UNet2DConditionModel
# with CLIP cross-attention and AutoencoderKL, both un-vendored): a 4-channel 64x64 latent denoiser with SD's widths
# (320 x (1,2,4,4), 2 ResBlocks per level, attention at 32/16/8 -- the text enters through the time embedding only, as
# in IF64_STANDIN) and the SD autoencoder's decoder geometry (128 x (1,2,4,4), 2+1 ResBlocks per level, one mid
# attention at 64x64 = 4096 tokens, 4 -> 3 channels, 64 -> 512 pixels)
SD64_STANDIN = UNetConfig(resolution=64, in_channels=4, out_ch=4, ch=320, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                          attn_resolutions=(32, 16, 8), gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=False)
SD_VAE_DECODER = UNetConfig(resolution=64, in_channels=4, out_ch=3, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                            attn_resolutions=(), gn_eps=1e-6, arch="dec")
# the encoder half of the same autoencoder (`vae.encode`, edit.py:594-597): 3 x 512 x 512 image -> 8 x 64 x 64 moments
SD_VAE_ENCODER = UNetConfig(resolution=512, in_channels=3, out_ch=8, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                            attn_resolutions=(), gn_eps=1e-6, arch="enc")
TINY_ENCODER = UNetConfig(resolution=64, in_channels=3, out_ch=8, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1,
                          attn_resolutions=(), gn_eps=1e-6, arch="enc")
TINY_LATENT = UNetConfig(resolution=16, in_channels=4, out_ch=4, ch=32, ch_mult=(1, 2), num_res_blocks=1,
                         attn_resolutions=(8,), gn_eps=1e-5, arch="adm", num_head_channels=16, learn_sigma=False)
TINY_DECODER = UNetConfig(resolution=16, in_channels=4, out_ch=3, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1,
                          attn_resolutions=(), gn_eps=1e-6, arch="dec")
# the same with text cross-attention behind every attention block (77 x 768 = the CLIP states of SD v1.x; 7 x 16 tiny)
SD64_XATTN_STANDIN = UNetConfig(resolution=64, in_channels=4, out_ch=4, ch=320, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                                attn_resolutions=(32, 16, 8), gn_eps=1e-5, arch="adm", num_head_channels=64,
                                learn_sigma=False, context_dim=768, context_len=77)
TINY_LATENT_XATTN = UNetConfig(resolution=16, in_channels=4, out_ch=4, ch=32, ch_mult=(1, 2), num_res_blocks=1,
                               attn_resolutions=(16, 8), gn_eps=1e-5, arch="adm", num_head_channels=16, learn_sigma=False,
                               context_dim=16, context_len=7)
# IF-I reads its T5 states (77 x 4096) through attention as well: the same shape class with cross-attention stages
IF64_XATTN_STANDIN = UNetConfig(resolution=64, ch=192, ch_mult=(1, 2, 3, 4), num_res_blocks=3, attn_resolutions=(32, 16, 8),
                                gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=True, context_dim=4096,
                                context_len=77)
TINY_ADM_XATTN = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,),
                            gn_eps=1e-5, arch="adm", num_head_channels=16, learn_sigma=True, context_dim=16, context_len=7)
TINY_ADM = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,),
                      gn_eps=1e-5, arch="adm", num_head_channels=16, learn_sigma=True)
# the same guided-diffusion UNetModel with `use_scale_shift_norm=False, resblock_updown=False` (additive embedding, conv
# down / up-sampling): the skeleton of the latent-diffusion denoiser, pinned against the reference's own class
TINY_ADM_PLAIN = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,),
                            gn_eps=1e-5, arch="adm", num_head_channels=16, learn_sigma=True, scale_shift_norm=False,
                            resblock_updown=False)
# 64-channel heads over 1024 and 256 tokens on a network the CPU oracle differentiates in seconds (flash attention tests)
FLASH_ADM = UNetConfig(resolution=32, ch=64, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(32, 16), gn_eps=1e-5,
                       arch="adm", num_head_channels=64, learn_sigma=True)
# Stable Diffusion v1.x denoiser (latent-diffusion UNetModel: 320 x (1,2,4,4), 2 ResBlocks per level, SpatialTransformer
# with 8 heads at the 64 / 32 / 16 latent resolutions and in the middle block, 77 x 768 CLIP states) and a small instance
SD15_UNET = UNetConfig(resolution=64, in_channels=4, out_ch=4, ch=320, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                       attn_resolutions=(64, 32, 16), gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=768,
                       context_len=77, scale_shift_norm=False, resblock_updown=False, num_heads=8, transformer_depth=1)
# Stable Diffusion 2.1-base (the id the shipped T-LOCO scripts name: scripts/main_T2I_StableDiffusion_null_space_projection*.sh:4,
# "stabilityai/stable-diffusion-2-1-base"): the same skeleton with 1024-wide OpenCLIP states, heads of 64 channels at
# every level (5 / 10 / 20 / 20 heads) and nn.Linear proj_in / proj_out (the 1x1 operator on [C][T]; checkpoints.py
# reshapes those weights); 865 910 724 parameters
SD21_BASE_UNET = UNetConfig(resolution=64, in_channels=4, out_ch=4, ch=320, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                            attn_resolutions=(64, 32, 16), gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=1024,
                            context_len=77, scale_shift_norm=False, resblock_updown=False, num_head_channels=64,
                            transformer_depth=1)
TINY_LDM = UNetConfig(resolution=16, in_channels=4, out_ch=4, ch=32, ch_mult=(1, 2), num_res_blocks=1,
                      attn_resolutions=(16, 8), gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=16, context_len=7,
                      scale_shift_norm=False, resblock_updown=False, num_heads=4, transformer_depth=1)
# a SpatialTransformer with 40-channel heads (Stable Diffusion's width at its 4096-token level) over 256 tokens
FLASH_LDM = UNetConfig(resolution=16, in_channels=4, out_ch=4, ch=160, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(16,),
                       gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=16, context_len=7, scale_shift_norm=False,
                       resblock_updown=False, num_heads=4, transformer_depth=1)
# a SpatialTransformer with 80-channel heads (Stable Diffusion v1's width at its 1024-token level) over 256 tokens: the
# three-channel-tile form of the flash attention kernels
FLASH_LDM80 = UNetConfig(resolution=16, in_channels=4, out_ch=4, ch=320, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(16,),
                         gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=16, context_len=7, scale_shift_norm=False,
                         resblock_updown=False, num_heads=4, transformer_depth=1)
# one level of Stable Diffusion's own width (320 channels, 8 heads of 40) over 64 tokens: the register-resident LayerNorm
# kernels (C = 64 * 5) and the 1-tap tile choice of the transformer's linear layers, at a size autodiff finishes in seconds
WIDE_LDM = UNetConfig(resolution=8, in_channels=4, out_ch=4, ch=320, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(8,),
                      gn_eps=1e-5, arch="adm", learn_sigma=False, context_dim=16, context_len=7, scale_shift_norm=False,
                      resblock_updown=False, num_heads=8, transformer_depth=1)
# config 5's geometry (64x64, four levels, attention at 32 / 16 / 8 incl. the 1024-token level, 64-channel heads, learned
# variance) at a third of IF64_STANDIN's width: the size the CPU reference solves in minutes (tests/golden/tloco_mid.pt)
MID_IF64 = UNetConfig(resolution=64, ch=64, ch_mult=(1, 2, 3, 4), num_res_blocks=2, attn_resolutions=(32, 16, 8),
                      gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=True)


# BASELINE config 5 on the architecture the shipped script names (scripts/main_T2I_DeepFloydIF_null_space_projection.sh:4,
# "DeepFloyd/IF-I-M-v1.0"): the stage-I U-Net of DeepFloyd IF = the guided-diffusion skeleton at 192 x (1,2,3,4) with 3
# ResBlocks per level, scale-shift norm, ResBlock resampling, 64-channel heads at 32 / 16 / 8, learned variance -- plus exact
# GELU, (skip + h) / sqrt 2, and attention over [text ; image] keys with the 77 x 4096 T5-XXL states projected to 768
IF_I_M_UNET = UNetConfig(resolution=64, ch=192, ch_mult=(1, 2, 3, 4), num_res_blocks=3, attn_resolutions=(32, 16, 8),
                         gn_eps=1e-5, arch="adm", num_head_channels=64, learn_sigma=True, context_dim=768, context_len=77,
                         act="gelu", res_scale=0.7071067811865476, added_kv=True, encoder_dim=4096)
# the other published stage-I sizes share the tree: IF-I-L at 320 channels (0.9 B parameters), IF-I-XL at 704 (4.3 B; one
# engine context keeps six layouts of every conv operator, so three XL contexts do not fit one GPU -- stated, not built for)
IF_I_WIDTH = {"M": 192, "L": 320, "XL": 704}


def if_stage1_config(size: str) -> UNetConfig:
    """`DeepFloyd/IF-I-<size>-v1.0` -> the stage-I U-Net configuration (`IF_I_M_UNET` with that size's width; the text states
    are projected to 4 * width channels)."""
    if size not in IF_I_WIDTH:
        raise ValueError(f"unknown DeepFloyd IF stage-I size {size!r} (known: {sorted(IF_I_WIDTH)})")
    ch = IF_I_WIDTH[size]
    return IF_I_M_UNET if size == "M" else UNetConfig(**{**IF_I_M_UNET.__dict__, "ch": ch, "context_dim": 4 * ch})


# the same switches at sizes autodiff on the CPU finishes in seconds (two attention levels, a resampling ResBlock each way,
# a channel-changing shortcut, 16-channel heads, 7 text states of width 24 -> 32)
TINY_IF = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16, 8), gn_eps=1e-5,
                     arch="adm", num_head_channels=16, learn_sigma=True, context_dim=32, context_len=7, act="gelu",
                     res_scale=0.7071067811865476, added_kv=True, encoder_dim=24)
# config 5's geometry (64 x 64, four levels, the 1024-token attention level, 64-channel heads) at a third of the width
MID_IF = UNetConfig(resolution=64, ch=64, ch_mult=(1, 2, 3, 4), num_res_blocks=2, attn_resolutions=(32, 16, 8), gn_eps=1e-5,
                    arch="adm", num_head_channels=64, learn_sigma=True, context_dim=256, context_len=77, act="gelu",
                    res_scale=0.7071067811865476, added_kv=True, encoder_dim=128)


def adm_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict layout of guided_diffusion ``UNetModel`` (unet.py:398-617) for the P2 flavour:
    ``time_embed.{0,2}``, ``input_blocks.N.{0,1}``, ``middle_block.{0,1,2}``, ``output_blocks.N.{0,1,2}``, ``out.{0,2}``;
    ResBlock = ``in_layers.{0,2}``, ``emb_layers.1``, ``out_layers.{0,3}``, ``skip_connection``;
    AttentionBlock = ``norm``, ``qkv`` (Conv1d), ``proj_out`` (Conv1d)."""
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    mc, ted = cfg.ch, cfg.ch * 4
    out_channels = cfg.out_ch * (2 if cfg.learn_sigma else 1)

    ecout = 2 if cfg.scale_shift_norm else 1

    def res(name, cin, cout):
        shapes[name + ".in_layers.0.weight"] = (cin,); shapes[name + ".in_layers.0.bias"] = (cin,)
        shapes[name + ".in_layers.2.weight"] = (cout, cin, 3, 3); shapes[name + ".in_layers.2.bias"] = (cout,)
        shapes[name + ".emb_layers.1.weight"] = (ecout * cout, ted); shapes[name + ".emb_layers.1.bias"] = (ecout * cout,)
        shapes[name + ".out_layers.0.weight"] = (cout,); shapes[name + ".out_layers.0.bias"] = (cout,)
        shapes[name + ".out_layers.3.weight"] = (cout, cout, 3, 3); shapes[name + ".out_layers.3.bias"] = (cout,)
        if cin != cout:
            shapes[name + ".skip_connection.weight"] = (cout, cin, 1, 1); shapes[name + ".skip_connection.bias"] = (cout,)

    def xfmr(name, c):
        """SpatialTransformer of latent-diffusion (ldm/modules/attention.py), depth 1."""
        D = cfg.context_dim
        shapes[name + ".norm.weight"] = (c,); shapes[name + ".norm.bias"] = (c,)
        shapes[name + ".proj_in.weight"] = (c, c, 1, 1); shapes[name + ".proj_in.bias"] = (c,)
        b = name + ".transformer_blocks.0"
        for n in ("norm1", "norm2", "norm3"):
            shapes[f"{b}.{n}.weight"] = (c,); shapes[f"{b}.{n}.bias"] = (c,)
        for n in ("to_q", "to_k", "to_v"):
            shapes[f"{b}.attn1.{n}.weight"] = (c, c)
        shapes[f"{b}.attn1.to_out.0.weight"] = (c, c); shapes[f"{b}.attn1.to_out.0.bias"] = (c,)
        shapes[f"{b}.attn2.to_q.weight"] = (c, c)
        shapes[f"{b}.attn2.to_k.weight"] = (c, D); shapes[f"{b}.attn2.to_v.weight"] = (c, D)
        shapes[f"{b}.attn2.to_out.0.weight"] = (c, c); shapes[f"{b}.attn2.to_out.0.bias"] = (c,)
        shapes[f"{b}.ff.net.0.proj.weight"] = (8 * c, c); shapes[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
        shapes[f"{b}.ff.net.2.weight"] = (c, 4 * c); shapes[f"{b}.ff.net.2.bias"] = (c,)
        shapes[name + ".proj_out.weight"] = (c, c, 1, 1); shapes[name + ".proj_out.bias"] = (c,)

    def attn(name, c):
        if cfg.transformer_depth > 0:
            return xfmr(name, c)
        shapes[name + ".norm.weight"] = (c,); shapes[name + ".norm.bias"] = (c,)
        shapes[name + ".qkv.weight"] = (3 * c, c, 1); shapes[name + ".qkv.bias"] = (3 * c,)
        shapes[name + ".proj_out.weight"] = (c, c, 1); shapes[name + ".proj_out.bias"] = (c,)
        if cfg.added_kv:            # DeepFloyd-IF AttentionBlock: GroupNorm + Conv1d of the text states, per head [k_h | v_h]
            shapes[name + ".norm_encoder.weight"] = (cfg.context_dim,); shapes[name + ".norm_encoder.bias"] = (cfg.context_dim,)
            shapes[name + ".encoder_kv.weight"] = (2 * c, cfg.context_dim, 1); shapes[name + ".encoder_kv.bias"] = (2 * c,)
            return
        if cfg.context_dim > 0:     # cross-attention stage (not part of guided_diffusion: stand-in for the diffusers blocks)
            x = name + ".xattn"
            shapes[x + ".norm.weight"] = (c,); shapes[x + ".norm.bias"] = (c,)
            shapes[x + ".q.weight"] = (c, c, 1); shapes[x + ".q.bias"] = (c,)
            shapes[x + ".k.weight"] = (c, cfg.context_dim); shapes[x + ".k.bias"] = (c,)
            shapes[x + ".v.weight"] = (c, cfg.context_dim); shapes[x + ".v.bias"] = (c,)
            shapes[x + ".proj_out.weight"] = (c, c, 1); shapes[x + ".proj_out.bias"] = (c,)

    shapes["time_embed.0.weight"] = (ted, mc); shapes["time_embed.0.bias"] = (ted,)
    shapes["time_embed.2.weight"] = (ted, ted); shapes["time_embed.2.bias"] = (ted,)
    if cfg.encoder_dim > 0:
        # host-side text conditioning of the IF U-Net (tloco.IFTextConditioner; diffusers `encoder_hid_proj` and
        # `add_embedding` = TextTimeEmbedding: LayerNorm, AttentionPooling, Linear, LayerNorm)
        E = cfg.encoder_dim
        shapes["encoder_proj.weight"] = (cfg.context_dim, E); shapes["encoder_proj.bias"] = (cfg.context_dim,)
        shapes["encoder_pooling.0.weight"] = (E,); shapes["encoder_pooling.0.bias"] = (E,)
        shapes["encoder_pooling.1.positional_embedding"] = (1, E)
        for n in ("k_proj", "q_proj", "v_proj"):
            shapes[f"encoder_pooling.1.{n}.weight"] = (E, E); shapes[f"encoder_pooling.1.{n}.bias"] = (E,)
        shapes["encoder_pooling.2.weight"] = (ted, E); shapes["encoder_pooling.2.bias"] = (ted,)
        shapes["encoder_pooling.3.weight"] = (ted,); shapes["encoder_pooling.3.bias"] = (ted,)
    ch = mc * cfg.ch_mult[0]
    shapes["input_blocks.0.0.weight"] = (ch, cfg.in_channels, 3, 3); shapes["input_blocks.0.0.bias"] = (ch,)
    chans = [ch]
    res_px = cfg.resolution
    ib = 1
    for lvl, mult in enumerate(cfg.ch_mult):
        for _ in range(cfg.num_res_blocks):
            res(f"input_blocks.{ib}.0", ch, mc * mult)
            ch = mc * mult
            if res_px in cfg.attn_resolutions:
                attn(f"input_blocks.{ib}.1", ch)
            chans.append(ch); ib += 1
        if lvl != len(cfg.ch_mult) - 1:
            if cfg.resblock_updown:
                res(f"input_blocks.{ib}.0", ch, ch)      # ResBlock(down=True)
            else:                                        # Downsample(ch, use_conv=True): conv3 stride 2 pad 1, unet.py:113-142
                shapes[f"input_blocks.{ib}.0.op.weight"] = (ch, ch, 3, 3); shapes[f"input_blocks.{ib}.0.op.bias"] = (ch,)
            chans.append(ch); ib += 1
            res_px //= 2
    res("middle_block.0", ch, ch); attn("middle_block.1", ch); res("middle_block.2", ch, ch)
    ob = 0
    for lvl, mult in list(enumerate(cfg.ch_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            res(f"output_blocks.{ob}.0", ch + ich, mc * mult)
            ch = mc * mult
            j = 1
            if res_px in cfg.attn_resolutions:
                attn(f"output_blocks.{ob}.{j}", ch); j += 1
            if lvl and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    res(f"output_blocks.{ob}.{j}", ch, ch)   # ResBlock(up=True)
                else:                                        # Upsample(ch, use_conv=True): nearest x2 + conv3, unet.py:83-110
                    shapes[f"output_blocks.{ob}.{j}.conv.weight"] = (ch, ch, 3, 3); shapes[f"output_blocks.{ob}.{j}.conv.bias"] = (ch,)
                res_px *= 2
            ob += 1
    shapes["out.0.weight"] = (ch,); shapes["out.0.bias"] = (ch,)
    shapes["out.2.weight"] = (out_channels, mc * cfg.ch_mult[0], 3, 3); shapes["out.2.bias"] = (out_channels,)
    return shapes


def enc_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict layout of the latent-diffusion ``Encoder`` (the network behind ``vae.encode`` of the reference's Stable
    Diffusion inversion, edit.py:594-597): ``conv_in``, ``down.L.block.B.{norm1,conv1,norm2,conv2,nin_shortcut}``,
    ``down.L.downsample.conv`` (stride 2 behind a (0,1,0,1) pad), ``mid.{block_1,attn_1,block_2}``, ``norm_out``, ``conv_out``
    (2 * z channels: mean and log-variance), followed by the autoencoder's 1x1 ``quant_conv``.  ``out_ch`` = 2 * z channels."""
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k); shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,); shapes[name + ".bias"] = (c,)

    def resblock(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    ch, mult = cfg.ch, tuple(cfg.ch_mult)
    in_mult = (1,) + mult
    conv("conv_in", cfg.in_channels, ch, 3)
    block_in = ch
    for lvl in range(len(mult)):
        block_in, block_out = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(cfg.num_res_blocks):
            resblock(f"down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            conv(f"down.{lvl}.downsample.conv", block_in, block_in, 3)
    resblock("mid.block_1", block_in, block_in)
    norm("mid.attn_1.norm", block_in)
    for q in ("q", "k", "v", "proj_out"):
        conv("mid.attn_1." + q, block_in, block_in, 1)
    resblock("mid.block_2", block_in, block_in)
    norm("norm_out", block_in)
    conv("conv_out", block_in, cfg.out_ch, 3)
    conv("quant_conv", cfg.out_ch, cfg.out_ch, 1)
    return shapes


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map following the constructor order of the
    reference module tree (``diffusion.py:41-126``)."""
    if cfg.arch == "adm":
        return adm_param_shapes(cfg)
    if cfg.arch == "dec":
        return dec_param_shapes(cfg)
    if cfg.arch == "enc":
        return enc_param_shapes(cfg)
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def lin(name, cin, cout):
        shapes[name + ".weight"] = (cout, cin)
        shapes[name + ".bias"] = (cout,)

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def resblock(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        lin(name + ".temb_proj", cfg.temb_ch, cout)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for p in ("q", "k", "v", "proj_out"):
            conv(name + "." + p, c, c, 1)

    ch, mult = cfg.ch, tuple(cfg.ch_mult)
    nres = len(mult)
    lin("temb.dense.0", ch, cfg.temb_ch)
    lin("temb.dense.1", cfg.temb_ch, cfg.temb_ch)
    conv("conv_in", cfg.in_channels, ch, 3)
    in_mult = (1,) + mult
    res = cfg.resolution
    block_in = ch
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        for b in range(cfg.num_res_blocks):
            resblock(f"down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if res in cfg.attn_resolutions:
                attn(f"down.{lvl}.attn.{b}", block_in)
        if lvl != nres - 1:
            conv(f"down.{lvl}.downsample.conv", block_in, block_in, 3)
            res //= 2
    resblock("mid.block_1", block_in, block_in)
    attn("mid.attn_1", block_in)
    resblock("mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * mult[lvl]
        skip_in = ch * mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            if b == cfg.num_res_blocks:
                skip_in = ch * in_mult[lvl]
            resblock(f"up.{lvl}.block.{b}", block_in + skip_in, block_out)
            block_in = block_out
            if res in cfg.attn_resolutions:
                attn(f"up.{lvl}.attn.{b}", block_in)
        if lvl != 0:
            conv(f"up.{lvl}.upsample.conv", block_in, block_in, 3)
            res *= 2
    norm("norm_out", block_in)
    conv("conv_out", block_in, cfg.out_ch, 3)
    return shapes


def dec_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict layout of the latent-diffusion ``Decoder`` (the module tree of the DDPM U-Net's up half without skip
    inputs and without ``temb_proj``): ``conv_in``, ``mid.{block_1,attn_1,block_2}``,
    ``up.L.block.B.{norm1,conv1,norm2,conv2,nin_shortcut}``, ``up.L.attn.B``, ``up.L.upsample.conv``, ``norm_out``,
    ``conv_out``, preceded by the autoencoder's ``post_quant_conv`` (1x1 on the latent channels)."""
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k); shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,); shapes[name + ".bias"] = (c,)

    def resblock(name, cin, cout):
        norm(name + ".norm1", cin); conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for p in ("q", "k", "v", "proj_out"):
            conv(name + "." + p, c, c, 1)

    ch, mult = cfg.ch, tuple(cfg.ch_mult)
    nlev = len(mult)
    block_in = ch * mult[-1]
    res = cfg.resolution
    conv("post_quant_conv", cfg.in_channels, cfg.in_channels, 1)      # AutoencoderKL.decode = decoder(post_quant_conv(z))
    conv("conv_in", cfg.in_channels, block_in, 3)
    resblock("mid.block_1", block_in, block_in); attn("mid.attn_1", block_in); resblock("mid.block_2", block_in, block_in)
    for lvl in reversed(range(nlev)):
        block_out = ch * mult[lvl]
        for b in range(cfg.num_res_blocks + 1):
            resblock(f"up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if res in cfg.attn_resolutions:
                attn(f"up.{lvl}.attn.{b}", block_in)
        if lvl != 0:
            conv(f"up.{lvl}.upsample.conv", block_in, block_in, 3)
            res *= 2
    norm("norm_out", block_in)
    conv("conv_out", block_in, cfg.out_ch, 3)
    return shapes


def _synth_tensor(cfg: UNetConfig, seed: int, name: str, shape) -> np.ndarray:
    if cfg.encoder_dim >= 1024 and name.startswith("encoder_pooling.1.") and len(shape) == 2 and shape[0] == shape[1]:
        # the three E x E maps of the attention pooling (3 x 16.8 M values at E = 4096): a cheap deterministic fill
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        row = rng.standard_normal(shape[1]).astype(np.float32) / np.sqrt(shape[1])
        return np.ascontiguousarray(np.stack([np.roll(row, i) for i in range(shape[0])]).astype(np.float32))
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    z = rng.standard_normal(shape).astype(np.float32)
    if name.endswith(".bias"):
        if (".norm" in name or name.startswith("norm_out") or name.endswith("in_layers.0.bias")
                or name.endswith("out_layers.0.bias") or name == "out.0.bias"):
            val = 0.1 * z
        else:
            val = 0.05 * z
    elif len(shape) == 1:  # norm gain
        val = 1.0 + 0.1 * z
    else:
        fan_in = int(np.prod(shape[1:]))
        val = z / np.sqrt(fan_in)
    return np.ascontiguousarray(val.astype(np.float32))


_SYNTH_CACHE: "OrderedDict[tuple, Dict[str, np.ndarray]]" = OrderedDict()
_SYNTH_CACHE_MAX_BYTES = 8 << 30


def synth_params(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic synthetic checkpoint (no hub / cluster weights offline).

    Every tensor is drawn from its own PCG64 stream keyed by (seed, crc32(name))
    so the values do not depend on iteration order or the torch version.  Scales
    keep activations O(1): conv/linear weights ~ N(0, 1/fan_in), norm gains
    1 + 0.1 N, biases 0.05 N.  No tensor is zero (a zero-initialised output
    conv would make d eps / d x vanish).

    The tensors are independent streams, so they are drawn on a thread pool (numpy releases the GIL while it fills), and the
    most recent results are kept per process (the at-size tests and the T-LOCO classes ask for the same 860 M-parameter set
    several times: 37 s each on 8 cores, three quarters of the former at-size test time).  The returned dict is a fresh one
    over shared arrays: treat the arrays as read-only.
    """
    key = (cfg, int(seed))
    hit = _SYNTH_CACHE.get(key)
    if hit is not None:
        _SYNTH_CACHE.move_to_end(key)
        return dict(hit)
    shapes = param_shapes(cfg)
    names = sorted(shapes, key=lambda n: -int(np.prod(shapes[n])))
    total = sum(int(np.prod(s)) for s in shapes.values())
    if total > (1 << 22):
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            vals = list(ex.map(lambda n: _synth_tensor(cfg, seed, n, shapes[n]), names))
        drawn = dict(zip(names, vals))
    else:
        drawn = {n: _synth_tensor(cfg, seed, n, shapes[n]) for n in names}
    out: Dict[str, np.ndarray] = {n: drawn[n] for n in shapes}      # the parameter list's own order
    _SYNTH_CACHE[key] = out
    size = lambda d: sum(v.nbytes for v in d.values())
    while len(_SYNTH_CACHE) > 1 and sum(size(d) for d in _SYNTH_CACHE.values()) > _SYNTH_CACHE_MAX_BYTES:
        _SYNTH_CACHE.popitem(last=False)
    return dict(out)
