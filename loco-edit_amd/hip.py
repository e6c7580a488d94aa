"""ctypes binding of ``libloco_hip.so`` (C ABI: ``include/loco_hip.h``).

PyTorch is used only as the owner of device memory and the current HIP stream;
every numerical operation below is a call into the hand-written gfx950 kernels.
There is NO fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from .config import UNetConfig

# LOCO_HIP_LIB: alternative build of the same library (A/B timing of kernel variants); default = the in-tree build
_LIB_PATH = os.environ.get("LOCO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libloco_hip.so")
_lib = None

# diagnostics of include/loco_hip_diag.h: only in a -DLOCO_DIAG build (make -C loco-edit_amd/csrc diag)
DIAG_SYMBOLS = ["loco_bench_conv", "loco_debug_tensor"]

# every symbol include/loco_hip.h declares
SYMBOLS = [
    "loco_version", "loco_device_count", "loco_create", "loco_fork", "loco_destroy", "loco_last_error",
    "loco_load_param", "loco_params_missing", "loco_unet_forward", "loco_ddim_step", "loco_sched_step",
    "loco_pmp_primal", "loco_pmp_set_second_mask", "loco_pmp_jvp", "loco_pmp_vjp", "loco_orthonormalize", "loco_qr_rows",
    "loco_convergence", "loco_convergence_rows", "loco_null_project", "loco_edit_axpy", "loco_mask_gather", "loco_mask_count",
    "loco_unet_flops", "loco_workspace_bytes", "loco_clock_stamp", "loco_set_side_stream", "loco_timer_start", "loco_timer_stop",
    "loco_profile_enable", "loco_profile_report", "loco_set_precision", "loco_get_precision", "loco_set_streams", "loco_set_chip_share",
    "loco_set_cond", "loco_set_context", "loco_lincomb", "loco_masked_axpby", "loco_latent_sample",
]


class LocoCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("resolution", C.c_int32), ("in_channels", C.c_int32), ("out_ch", C.c_int32), ("ch", C.c_int32),
        ("num_levels", C.c_int32), ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32),
        ("num_attn_res", C.c_int32), ("attn_resolutions", C.c_int32 * 8), ("gn_groups", C.c_int32),
        ("gn_eps", C.c_float), ("max_batch", C.c_int32),
        ("arch", C.c_int32), ("num_head_channels", C.c_int32), ("learn_sigma", C.c_int32),
        ("context_dim", C.c_int32), ("context_len", C.c_int32),
        ("scale_shift_norm", C.c_int32), ("resblock_updown", C.c_int32), ("num_heads", C.c_int32),
        ("transformer_depth", C.c_int32),
        ("act", C.c_int32), ("res_scale", C.c_float), ("added_kv", C.c_int32),
    ]


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the engine and declare prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.loco_version.restype = C.c_char_p
    lib.loco_device_count.restype = C.c_int
    lib.loco_create.argtypes = [C.POINTER(LocoCfg), C.POINTER(vp)]
    if hasattr(lib, "loco_fork"):      # (a library of an earlier round, loaded by LOCO_HIP_LIB for an A/B: everything but fork works)
        lib.loco_fork.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    lib.loco_destroy.argtypes = [vp]
    lib.loco_destroy.restype = None
    lib.loco_last_error.argtypes = [vp]
    lib.loco_last_error.restype = C.c_char_p
    lib.loco_load_param.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32]
    lib.loco_params_missing.argtypes = [vp]
    lib.loco_unet_forward.argtypes = [vp, vp, f32, i32, vp, vp]
    lib.loco_ddim_step.argtypes = [vp, vp, f32, f32, f32, f32, vp, i32, vp, vp]
    lib.loco_sched_step.argtypes = [vp, vp, vp, f32, f32, f32, vp, i64, vp, vp, vp]
    lib.loco_pmp_primal.argtypes = [vp, vp, f32, f32, vp, i32, vp]
    lib.loco_pmp_set_second_mask.argtypes = [vp, vp, i32, vp]
    lib.loco_pmp_jvp.argtypes = [vp, vp, i32, vp, vp]
    lib.loco_pmp_vjp.argtypes = [vp, vp, i32, vp, vp]
    lib.loco_orthonormalize.argtypes = [vp, vp, i32, i64, vp, vp]
    lib.loco_qr_rows.argtypes = [vp, vp, i32, i64, vp]
    lib.loco_convergence.argtypes = [vp, vp, vp, i64, f32, vp, vp]
    lib.loco_convergence_rows.argtypes = [vp, vp, vp, i32, i64, f32, vp, vp]
    lib.loco_null_project.argtypes = [vp, vp, i32, vp, i32, i64, vp, vp]
    lib.loco_edit_axpy.argtypes = [vp, vp, vp, C.POINTER(f32), i32, i64, vp, vp]
    lib.loco_mask_gather.argtypes = [vp, vp, i32, vp, vp]
    lib.loco_mask_count.argtypes = [vp]
    lib.loco_mask_count.restype = i64
    lib.loco_clock_stamp.argtypes = [vp, vp, vp]
    lib.loco_set_side_stream.argtypes = [vp, vp]
    lib.loco_unet_flops.argtypes = [vp]
    lib.loco_unet_flops.restype = C.c_double
    lib.loco_workspace_bytes.argtypes = [vp]
    lib.loco_workspace_bytes.restype = i64
    lib.loco_timer_start.argtypes = [vp, vp]
    lib.loco_timer_stop.argtypes = [vp, vp, C.POINTER(f32)]
    lib.loco_set_precision.argtypes = [vp, i32]
    lib.loco_get_precision.argtypes = [vp]
    lib.loco_set_streams.argtypes = [vp, i32]
    lib.loco_set_chip_share.argtypes = [vp, i32]
    lib.loco_set_cond.argtypes = [vp, vp, vp]
    lib.loco_set_context.argtypes = [vp, vp, vp]
    lib.loco_masked_axpby.argtypes = [vp, vp, vp, f32, f32, i32, vp, vp]
    lib.loco_latent_sample.argtypes = [vp, vp, vp, f32, i32, vp, vp]
    lib.loco_lincomb.argtypes = [vp, C.POINTER(vp), C.POINTER(f32), i32, vp, i64, vp]
    lib.loco_profile_enable.argtypes = [vp, i32]
    lib.loco_profile_report.argtypes = [vp, C.c_char_p, i64]
    if hasattr(lib, "loco_bench_conv"):          # diag build only
        lib.loco_bench_conv.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(f32), vp]
        lib.loco_debug_tensor.argtypes = [vp, C.c_char_p, vp, i64, vp]
        lib.loco_debug_tensor.restype = i64
    _lib = lib
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_dev(t: torch.Tensor, dtype=torch.float32):
    if not t.is_cuda:
        raise ValueError("loco_hip operates on device tensors only")
    if t.dtype != dtype:
        raise ValueError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")


class LocoEngine:
    """One engine (= loco_ctx) per process and GPU."""

    def __init__(self, cfg: UNetConfig, max_batch: int = 8, device: Optional[torch.device] = None):
        self.lib = load_library()
        if not torch.cuda.is_available() or self.lib.loco_device_count() < 1:
            raise RuntimeError("loco_hip: no HIP device visible; the hot path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:0")
        torch.cuda.set_device(self.device)
        self.cfg = cfg
        self.max_batch = int(max_batch)
        c = LocoCfg()
        c.struct_size = C.sizeof(LocoCfg)
        c.resolution, c.in_channels, c.out_ch, c.ch = cfg.resolution, cfg.in_channels, cfg.out_ch, cfg.ch
        c.num_levels = len(cfg.ch_mult)
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.num_res_blocks = cfg.num_res_blocks
        c.num_attn_res = len(cfg.attn_resolutions)
        for i, r in enumerate(cfg.attn_resolutions):
            c.attn_resolutions[i] = r
        c.gn_groups, c.gn_eps, c.max_batch = cfg.gn_groups, cfg.gn_eps, self.max_batch
        c.arch = {"ddpm": 0, "adm": 1, "dec": 2, "enc": 3}[cfg.arch]
        c.num_head_channels, c.learn_sigma = cfg.num_head_channels, int(cfg.learn_sigma)
        c.context_dim, c.context_len = cfg.context_dim, cfg.context_len
        c.scale_shift_norm, c.resblock_updown = int(cfg.scale_shift_norm), int(cfg.resblock_updown)
        c.num_heads, c.transformer_depth = cfg.num_heads, cfg.transformer_depth
        c.act, c.res_scale, c.added_kv = {"silu": 0, "gelu": 1}[cfg.act], float(cfg.res_scale), int(cfg.added_kv)
        self._ctx = C.c_void_p()
        rc = self.lib.loco_create(C.byref(c), C.byref(self._ctx))
        if rc != 0:
            msg = self.lib.loco_last_error(self._ctx).decode() if self._ctx else "?"
            raise RuntimeError(f"loco_create failed ({rc}): {msg}")
        self.n = cfg.n              # elements of the network input (image / latent)
        self.n_out = cfg.n_out      # elements of its output (= n for the denoisers; the decoded image for arch "dec")

    def fork(self, max_batch: Optional[int] = None) -> "LocoEngine":
        """A second engine context on THIS engine's parameters (loco_fork): shares the device copies of the weights in every
        layout, owns its arenas / statistics / scratch / per-prompt constants.  What the reference does with one U-Net object
        for all classifier-free-guidance branches (edit.py:1319-1322, :655-667); bit-identical to an independent engine."""
        child = object.__new__(LocoEngine)
        child.lib, child.cfg, child.device = self.lib, self.cfg, self.device
        child.max_batch = int(max_batch or self.max_batch)
        child.n, child.n_out = self.n, self.n_out
        for k, v in self.__dict__.items():      # host-side settings (stream mode ...) that __init__ derives from the arguments
            if k not in child.__dict__ and k != "_ctx":
                child.__dict__[k] = v
        child._ctx = C.c_void_p()
        torch.cuda.set_device(self.device)
        rc = self.lib.loco_fork(self._ctx, child.max_batch, C.byref(child._ctx))
        if rc != 0:
            msg = self.lib.loco_last_error(child._ctx).decode() if child._ctx else self.lib.loco_last_error(self._ctx).decode()
            raise RuntimeError(f"loco_fork failed ({rc}): {msg}")
        return child

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self.lib.loco_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.loco_last_error(self._ctx).decode()}")

    # ---- parameters (model.load_state_dict, reference utils.py:102-105)
    def load_state_dict(self, sd: Dict[str, "np.ndarray | torch.Tensor"]):
        for name, v in sd.items():
            if self.cfg.encoder_dim > 0 and name.startswith(("encoder_proj.", "encoder_pooling.")):
                continue        # the image-independent text conditioning of the IF U-Net lives on the host (tloco.IFTextConditioner)
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            rc = self.lib.loco_load_param(self._ctx, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, 0)
            self._check(rc, f"loco_load_param({name})")
        miss = self.lib.loco_params_missing(self._ctx)
        if miss != 0:
            raise RuntimeError(f"{miss} parameters missing: {self.lib.loco_last_error(self._ctx).decode()}")

    # ---- denoiser
    def _chk_input(self, x: torch.Tensor):
        """[B, in_channels, R, R] of this network (the C ABI takes a pointer and a count: a wrong shape would be read
        as garbage, not refused; the batch bound is checked there)."""
        want = (self.cfg.in_channels, self.cfg.resolution, self.cfg.resolution)
        if x.dim() != 4 or tuple(x.shape[1:]) != want:
            raise ValueError(f"input must be [B, {want[0]}, {want[1]}, {want[2]}], got {tuple(x.shape)}")

    def unet_forward(self, x: torch.Tensor, t: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        _chk_dev(x)
        self._chk_input(x)
        if out is not None:
            _chk_dev(out)
            if out.numel() != x.shape[0] * self.n_out:
                raise ValueError(f"out must hold {x.shape[0]} x {self.n_out} elements, got {tuple(out.shape)}")
            eps = out
        elif self.cfg.arch in ("dec", "enc"):      # decoder / encoder: [B, C_in, R, R] -> [B, out_ch, R_out, R_out]
            eps = torch.empty(x.shape[0], self.cfg.out_ch, self.cfg.out_resolution, self.cfg.out_resolution,
                              device=x.device, dtype=torch.float32)
        else:
            eps = torch.empty_like(x)
        self._check(self.lib.loco_unet_forward(self._ctx, _ptr(x), float(t), x.shape[0], _ptr(eps), _stream()),
                    "loco_unet_forward")
        return eps

    def ddim_step(self, x, t, at, at_next, eta=0.0, noise=None, out=None):
        _chk_dev(x)
        self._chk_input(x)
        if noise is not None:
            _chk_dev(noise)
        out = torch.empty_like(x) if out is None else out
        self._check(self.lib.loco_ddim_step(self._ctx, _ptr(x), float(t), float(at), float(at_next), float(eta),
                                            _ptr(noise), x.shape[0], _ptr(out), _stream()), "loco_ddim_step")
        return out

    def sched_step(self, x, et, at, at_next, eta=0.0, noise=None, want_x0=False):
        _chk_dev(x)
        _chk_dev(et)
        out = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        self._check(self.lib.loco_sched_step(self._ctx, _ptr(x), _ptr(et), float(at), float(at_next), float(eta),
                                             _ptr(noise), x.numel(), _ptr(out), _ptr(x0), _stream()),
                    "loco_sched_step")
        return out, x0

    # ---- PMP-Jacobian operator
    def pmp_primal(self, x, t, at, mask: Optional[torch.Tensor] = None, use_et: bool = False):
        _chk_dev(x)
        if x.numel() != self.n:
            raise ValueError(f"the linearisation point is one sample of {self.n} elements, got {tuple(x.shape)}")
        m8 = None
        if mask is not None:
            m8 = mask.to(device=x.device, dtype=torch.uint8).contiguous().view(-1)
            if m8.numel() != self.n_out:
                raise ValueError("mask must have C*H*W elements (of the network output)")
        self._mask_keepalive = m8
        self._check(self.lib.loco_pmp_primal(self._ctx, _ptr(x), float(t), float(at), _ptr(m8), int(use_et),
                                             _stream()), "loco_pmp_primal")

    def pmp_set_second_mask(self, mask2: Optional[torch.Tensor], from_row: int = 0):
        """Rows >= from_row of later pmp_jvp / pmp_vjp calls use ``mask2`` (None: off)."""
        m8 = None
        if mask2 is not None:
            m8 = mask2.to(device=self.device, dtype=torch.uint8).contiguous().view(-1)
            if m8.numel() != self.n_out:
                raise ValueError("mask must have C*H*W elements (of the network output)")
        self._mask2_keepalive = m8                   # the copy is enqueued on the stream: keep the source alive, no host sync
        self._check(self.lib.loco_pmp_set_second_mask(self._ctx, _ptr(m8), int(from_row), _stream()), "loco_pmp_set_second_mask")

    def pmp_jvp(self, V: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``out``: a [k, n_out] tensor to write into (callers that launch on a side stream allocate it on their own
        stream first, so no allocation happens under the side stream)."""
        _chk_dev(V)
        k = V.shape[0]
        U = torch.empty(k, self.n_out, device=V.device, dtype=torch.float32) if out is None else out
        if out is not None:
            _chk_dev(out)
            if tuple(out.shape) != (k, self.n_out):
                raise ValueError(f"out must be {(k, self.n_out)}, got {tuple(out.shape)}")
        self._check(self.lib.loco_pmp_jvp(self._ctx, _ptr(V), k, _ptr(U), _stream()), "loco_pmp_jvp")
        return U

    def pmp_vjp(self, U: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        _chk_dev(U)
        k = U.shape[0]
        A = torch.empty(k, self.n, device=U.device, dtype=torch.float32) if out is None else out
        if out is not None:
            _chk_dev(out)
            if tuple(out.shape) != (k, self.n):
                raise ValueError(f"out must be {(k, self.n)}, got {tuple(out.shape)}")
        self._check(self.lib.loco_pmp_vjp(self._ctx, _ptr(U), k, _ptr(A), _stream()), "loco_pmp_vjp")
        return A

    # ---- solver algebra
    def orthonormalize_(self, A: torch.Tensor) -> torch.Tensor:
        _chk_dev(A)
        k, n = A.shape
        s = torch.empty(k, device=A.device, dtype=torch.float32)
        self._check(self.lib.loco_orthonormalize(self._ctx, _ptr(A), k, n, _ptr(s), _stream()), "loco_orthonormalize")
        return s

    def qr_rows_(self, A: torch.Tensor):
        _chk_dev(A)
        k, n = A.shape
        self._check(self.lib.loco_qr_rows(self._ctx, _ptr(A), k, n, _stream()), "loco_qr_rows")
        return A

    def convergence(self, Vprev, V, atol) -> torch.Tensor:
        _chk_dev(Vprev)
        _chk_dev(V)
        out = torch.empty(2, device=V.device, dtype=torch.float32)
        self._check(self.lib.loco_convergence(self._ctx, _ptr(Vprev), _ptr(V), V.numel(), float(atol), _ptr(out),
                                              _stream()), "loco_convergence")
        return out

    def convergence_rows(self, Vprev, V, atol) -> torch.Tensor:
        """[distance, allclose flag] of the rows of V against +-the rows of Vprev (loco_convergence_rows)."""
        _chk_dev(Vprev)
        _chk_dev(V)
        k, n = V.shape
        out = torch.empty(2, device=V.device, dtype=torch.float32)
        self._check(self.lib.loco_convergence_rows(self._ctx, _ptr(Vprev), _ptr(V), k, n, float(atol), _ptr(out),
                                                   _stream()), "loco_convergence_rows")
        return out

    def null_project(self, Vm, Vn=None) -> torch.Tensor:
        _chk_dev(Vm)
        k, n = Vm.shape
        k0 = 0
        if Vn is not None:
            _chk_dev(Vn)
            k0 = Vn.shape[0]
        out = torch.empty_like(Vm)
        self._check(self.lib.loco_null_project(self._ctx, _ptr(Vm), k, _ptr(Vn), k0, n, _ptr(out), _stream()),
                    "loco_null_project")
        return out

    def edit_axpy(self, x, v, alphas) -> torch.Tensor:
        _chk_dev(x)
        _chk_dev(v)
        B = len(alphas)
        n = x.numel()
        out = torch.empty((B,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
        arr = (C.c_float * B)(*[float(a) for a in alphas])
        self._check(self.lib.loco_edit_axpy(self._ctx, _ptr(x), _ptr(v), arr, B, n, _ptr(out), _stream()),
                    "loco_edit_axpy")
        torch.cuda.current_stream().synchronize()   # `arr` is host memory read asynchronously
        return out

    def mask_gather(self, U) -> torch.Tensor:
        _chk_dev(U)
        k = U.shape[0]
        L = int(self.lib.loco_mask_count(self._ctx))
        out = torch.empty(k, L, device=U.device, dtype=torch.float32)
        self._check(self.lib.loco_mask_gather(self._ctx, _ptr(U), k, _ptr(out), _stream()), "loco_mask_gather")
        return out

    def mask_count(self) -> int:
        """L = number of selected elements of the mask given to the last ``pmp_primal`` (n when unmasked)."""
        return int(self.lib.loco_mask_count(self._ctx))

    # ---- conditioning / CFG combination (T-LOCO)
    def set_cond(self, emb_add: Optional[torch.Tensor]):
        """Conditioning embedding [4*ch] added to the time embedding before its SiLU (None clears it)."""
        if emb_add is not None:
            _chk_dev(emb_add)
            if emb_add.numel() != 4 * self.cfg.ch:
                raise ValueError("conditioning embedding must have 4*ch elements")
        self._check(self.lib.loco_set_cond(self._ctx, _ptr(emb_add), _stream()), "loco_set_cond")

    def set_context(self, tokens: torch.Tensor):
        """Encoder states of the prompt [context_len, context_dim] for the cross-attention stages (the
        ``encoder_hidden_states`` of ``self.unet(...)``, edit.py:664-667); projected to keys / values once."""
        _chk_dev(tokens)
        if tuple(tokens.shape) != (self.cfg.context_len, self.cfg.context_dim):
            raise ValueError(f"context must be [{self.cfg.context_len}, {self.cfg.context_dim}], got {tuple(tokens.shape)}")
        self._check(self.lib.loco_set_context(self._ctx, _ptr(tokens), _stream()), "loco_set_context")
        torch.cuda.current_stream().synchronize()     # `tokens` may be a temporary

    def masked_axpby(self, V: torch.Tensor, E: torch.Tensor, cv: float, ce: float) -> torch.Tensor:
        """mask * (cv*V + ce*E) with the mask of the last pmp_primal; V, E: [k, n]."""
        _chk_dev(V)
        _chk_dev(E)
        out = torch.empty_like(V)
        self._check(self.lib.loco_masked_axpby(self._ctx, _ptr(V), _ptr(E), float(cv), float(ce), V.shape[0], _ptr(out),
                                               _stream()), "loco_masked_axpby")
        return out

    def latent_sample(self, moments: torch.Tensor, noise: Optional[torch.Tensor], scale: float) -> torch.Tensor:
        """Encoder contexts: scale * (mean + std * noise) from the moments [B, 2Z, h, w] (noise None: the mean)."""
        _chk_dev(moments)
        B, C2, h, w = moments.shape
        if C2 * h * w != self.n_out:
            raise ValueError(f"moments must be [B, {self.cfg.out_ch}, ...] of this encoder, got {tuple(moments.shape)}")
        if noise is not None:
            _chk_dev(noise)
            if tuple(noise.shape) != (B, C2 // 2, h, w):
                raise ValueError(f"noise must be {(B, C2 // 2, h, w)}, got {tuple(noise.shape)}")
        z = torch.empty(B, C2 // 2, h, w, device=moments.device, dtype=torch.float32)
        self._check(self.lib.loco_latent_sample(self._ctx, _ptr(moments), _ptr(noise) if noise is not None else None,
                                                float(scale), B, _ptr(z), _stream()), "loco_latent_sample")
        return z

    def lincomb(self, terms, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sum_i coef_i * tensor_i for [(coef, tensor), ...] (<= 4 terms, same shape, fp32, contiguous)."""
        for _, t in terms:
            _chk_dev(t)
        out = torch.empty_like(terms[0][1]) if out is None else out
        n = len(terms)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in terms])
        coef = (C.c_float * n)(*[float(c) for c, _ in terms])
        self._check(self.lib.loco_lincomb(self._ctx, ptrs, coef, n, _ptr(out), out.numel(), _stream()), "loco_lincomb")
        return out

    # ---- introspection
    def version(self) -> str:
        return self.lib.loco_version().decode()

    def clock_stamp(self) -> torch.Tensor:
        """Enqueue a {s_memtime, s_memrealtime} stamp on the current stream; returns the device int64[2] it lands in."""
        out = torch.zeros(2, device=self.device, dtype=torch.int64)
        self._check(self.lib.loco_clock_stamp(self._ctx, _ptr(out), _stream()), "loco_clock_stamp")
        return out

    @staticmethod
    def sclk_mhz(stamp0: torch.Tensor, stamp1: torch.Tensor) -> float:
        """Average shader clock between two stamps (after a synchronize)."""
        d = (stamp1 - stamp0).tolist()
        return 100.0 * d[0] / max(d[1], 1)

    def unet_flops(self) -> float:
        return float(self.lib.loco_unet_flops(self._ctx))

    def workspace_bytes(self) -> int:
        return int(self.lib.loco_workspace_bytes(self._ctx))

    def timer_start(self):
        self._check(self.lib.loco_timer_start(self._ctx, _stream()), "loco_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.loco_timer_stop(self._ctx, _stream(), C.byref(ms)), "loco_timer_stop")
        return float(ms.value)

    PRECISIONS = {"f32": 0, "bf16x3": 1, "f16": 2}

    def set_precision(self, mode: str):
        """'f32' = exact fp32 MFMA (parity anchor); 'bf16x3' = split-bf16 MFMA (fp32-faithful to ~2^-16);
        'f16' = one f16 MFMA per product (11-bit operands, fp32 accumulate)."""
        self._check(self.lib.loco_set_precision(self._ctx, self.PRECISIONS[mode]), "loco_set_precision")

    def set_streams(self, n: int):
        """Probe groups of a tangent / cotangent pass on 1 (default) or 2 HIP streams (identical results)."""
        self._check(self.lib.loco_set_streams(self._ctx, int(n)), "loco_set_streams")

    def set_chip_share(self, n: int):
        """n engine contexts run their passes side by side on different streams (T-LOCO's guidance branches): split-K then aims
        at 256 / n workgroups per launch (include/loco_hip.h `loco_set_chip_share`)."""
        self._check(self.lib.loco_set_chip_share(self._ctx, int(n)), "loco_set_chip_share")

    def set_side_stream(self, stream: "Optional[torch.cuda.Stream]"):
        """The second stream of `set_streams(2)`: a stream the caller measured to run BESIDE its current stream (HIP hands
        hardware queues out round-robin; `tloco.BranchStreams._pick` does the measurement).  None: the context's own."""
        self._side_stream = stream            # keep the torch object alive while the context may enqueue on it
        ptr = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._check(self.lib.loco_set_side_stream(self._ctx, ptr), "loco_set_side_stream")

    def set_streams_measured(self, n: int):
        """`set_streams(n)`; for n = 2 the side stream is chosen by measurement (falls back to one stream when no stream of
        this process runs beside the current one)."""
        if n == 2:
            from .tloco import BranchStreams
            side = BranchStreams._pick(1, self.device)
            if not side:
                self.set_streams(1)
                return 1
            self.set_side_stream(side[0])
        self.set_streams(n)
        return n

    def get_precision(self) -> str:
        m = self.lib.loco_get_precision(self._ctx)
        return {v: k for k, v in self.PRECISIONS.items()}[m]

    def _need_diag(self, sym):
        if not hasattr(self.lib, sym):
            raise RuntimeError(f"{sym} is a diagnostic of include/loco_hip_diag.h: build `make -C loco-edit_amd/csrc diag` "
                               "and set LOCO_HIP_LIB=<repo>/loco-edit_amd/libloco_hip_diag.so")

    def bench_conv(self, cin, cout, H, W, B, mode, taps=9, tile=-1, iters=20) -> float:
        self._need_diag("loco_bench_conv")
        ms = C.c_float()
        self._check(self.lib.loco_bench_conv(self._ctx, cin, cout, H, W, B, mode, taps, tile, iters, C.byref(ms),
                                             _stream()), "loco_bench_conv")
        return float(ms.value)

    def profile_enable(self, on):
        """True/1: per kernel variant; 2: per layer shape; False: off."""
        self._check(self.lib.loco_profile_enable(self._ctx, int(on)), "loco_profile_enable")

    def profile_report(self):
        """-> {kernel variant: dict(launches, ms, flops)} for the conv launches since profile_enable(True)."""
        buf = C.create_string_buffer(1 << 18)
        self._check(self.lib.loco_profile_report(self._ctx, buf, len(buf)), "loco_profile_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, fl = line.rsplit(" ", 3)
            out[name] = dict(launches=int(float(n)), ms=float(ms), flops=float(fl))
        return out

    def debug_tensor(self, name: str, numel: int) -> torch.Tensor:
        self._need_diag("loco_debug_tensor")
        dst = torch.empty(numel, device=self.device, dtype=torch.float32)
        got = self.lib.loco_debug_tensor(self._ctx, name.encode(), _ptr(dst), numel, _stream())
        if got < 0:
            raise RuntimeError(f"loco_debug_tensor({name}) failed: {self.lib.loco_last_error(self._ctx).decode()}")
        return dst[:got]
