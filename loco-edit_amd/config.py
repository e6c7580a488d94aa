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

    @property
    def temb_ch(self) -> int:
        return self.ch * 4

    @property
    def n(self) -> int:
        return self.in_channels * self.resolution * self.resolution


# configs[0..1] of BASELINE.json: google/ddpm-celebahq-256 architecture
CELEBA_DDPM = UNetConfig()
# parity-test sizes the CPU oracle finishes in seconds
TINY_DDPM = UNetConfig(resolution=32, ch=32, ch_mult=(1, 2, 2), num_res_blocks=2,
                       attn_resolutions=(16,))
MID_DDPM = UNetConfig(resolution=64, ch=32, ch_mult=(1, 1, 2), num_res_blocks=1,
                      attn_resolutions=(16,))


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map following the constructor order of the
    reference module tree (``diffusion.py:41-126``)."""
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


def synth_params(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic synthetic checkpoint (no hub / cluster weights offline).

    Every tensor is drawn from its own PCG64 stream keyed by (seed, crc32(name))
    so the values do not depend on iteration order or the torch version.  Scales
    keep activations O(1): conv/linear weights ~ N(0, 1/fan_in), norm gains
    1 + 0.1 N, biases 0.05 N.  No tensor is zero (a zero-initialised output
    conv would make d eps / d x vanish).
    """
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg).items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        z = rng.standard_normal(shape).astype(np.float32)
        if name.endswith(".bias"):
            if ".norm" in name or name.startswith("norm_out"):
                val = 0.1 * z
            else:
                val = 0.05 * z
        elif len(shape) == 1:  # norm gain
            val = 1.0 + 0.1 * z
        else:
            fan_in = int(np.prod(shape[1:]))
            val = z / np.sqrt(fan_in)
        out[name] = np.ascontiguousarray(val.astype(np.float32))
    return out
