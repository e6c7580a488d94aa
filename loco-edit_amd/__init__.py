"""loco-edit_amd: MI355X-native hot path of LOCO-Edit (null-space projection).

Scope: DDIM inversion -> PMP-Jacobian low-rank subspace
solver -> null-space projection -> masked edit step -> DDIM decode, for the
unconditional DDPM denoiser.  All numerics run in hand-written HIP kernels for
gfx950 behind the C ABI declared in ``include/loco_hip.h``; this package is the
Python host side mirroring the reference's ``src/modules/edit.py`` interface.
"""
from .config import UNetConfig, CELEBA_DDPM, TINY_DDPM, MID_DDPM, FFHQ_P2, TINY_ADM  # noqa: F401
