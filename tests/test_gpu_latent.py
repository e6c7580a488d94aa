"""GPU tests of the latent (Stable Diffusion-shaped) path: the decoder network engine (arch "dec") against the torch
restatement in oracle/loco_oracle.py -- forward, J V (torch.func.jvp of the restatement) and U^T J (autograd) with the
mask on the decoded image -- in the exact-fp32 and the split-bf16 arithmetic.  Bars as in test_gpu_parity.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import loco_edit_amd  # noqa: E402,F401
import loco_oracle as orc  # noqa: E402
from loco_edit_amd.config import TINY_DECODER, synth_params  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"f32": 2e-5, "bf16x3": 2e-4}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_decoder_forward_jvp_vjp_vs_restatement(prec):
    from loco_edit_amd.hip import LocoEngine
    cfg = TINY_DECODER
    params = synth_params(cfg, 0)
    eng = LocoEngine(cfg, max_batch=4, device=torch.device(DEV))
    eng.load_state_dict(params)
    eng.set_precision(prec)
    p = orc.to_torch(params)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, cfg.in_channels, cfg.resolution, cfg.resolution, generator=g)
    R = cfg.out_resolution
    assert (eng.n, eng.n_out) == (4 * 16 * 16, 3 * 64 * 64) and R == 64
    x_ref = orc.decoder_forward(p, cfg, z)
    x = eng.unet_forward(z.to(DEV), 0.0)
    assert x.shape == (1, 3, R, R) and rel(x, x_ref) < TOL[prec]
    zb = torch.cat([z, 0.5 * z, z + 0.1], dim=0)                      # batch of 3 through one launch list
    assert rel(eng.unet_forward(zb.to(DEV), 0.0), orc.decoder_forward(p, cfg, zb)) < TOL[prec]
    with pytest.raises(RuntimeError):
        eng.ddim_step(z.to(DEV), 10.0, 0.5, 0.6)
    with pytest.raises(RuntimeError):
        eng.pmp_primal(z.to(DEV), 0.0, 1.0, None, use_et=False)     # no x0 combination for a decoder
    mask = torch.zeros(3, R, R, dtype=torch.bool); mask[:, 20:40, 10:50] = True
    V = torch.randn(3, eng.n, generator=g)
    f = lambda z_: orc.decoder_forward(p, cfg, z_)
    JV = torch.stack([torch.func.jvp(f, (z,), (v.view_as(z),))[1].reshape(-1) for v in V])
    for m in (None, mask):
        eng.pmp_primal(z.to(DEV), 0.0, 1.0, None if m is None else m.to(DEV), use_et=True)
        U = eng.pmp_jvp(V.to(DEV))
        ref = JV if m is None else JV * m.reshape(1, -1)
        assert U.shape == (3, eng.n_out) and rel(U, ref) < TOL[prec] * 5
        Uc = torch.randn(3, eng.n_out, generator=g)
        A = eng.pmp_vjp(Uc.to(DEV))
        zz = z.clone().requires_grad_(True)
        out = orc.decoder_forward(p, cfg, zz).reshape(-1)
        Aref = torch.stack([torch.autograd.grad((out * (u if m is None else u * m.reshape(-1))).sum(), zz,
                                                retain_graph=True)[0].reshape(-1) for u in Uc])
        assert A.shape == (3, eng.n) and rel(A, Aref) < TOL[prec] * 5
        # adjointness <J V, U> == <V, J^T U> on the device results
        lhs = (U.double().cpu() * (Uc.double() if m is None else Uc.double() * m.reshape(1, -1))).sum()
        rhs = (V.double() * A.double().cpu()).sum()
        assert abs(lhs - rhs) / abs(lhs) < 1e-4
        if m is not None:
            assert eng.mask_count() == int(m.sum())
            assert torch.equal(eng.mask_gather(U).cpu(), U.cpu()[:, m.reshape(-1)])
