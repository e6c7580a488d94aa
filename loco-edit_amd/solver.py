"""PMP-Jacobian low-rank subspace solver: block power iteration on J^T J with
J = d x0_hat[mask] / d x_t (reference ``local_encoder_decoder_pullback_xt``,
``src/modules/edit.py:2406-2504``).

The reference pays (3k+1) denoiser passes per iteration (jacfwd recomputes the
primal k times, k sequential backward sweeps) plus three host round trips; here
the primal is evaluated once per solve, each iteration is one batched tangent
pass + one batched cotangent pass + an on-device Gram/eig re-orthonormalisation,
and at most one 2-float readback.

Multi-GPU: probes are sharded over ranks (``ProbeSharder``), each
rank runs JVP+VJP on its rows, one all-gather of the A shards per iteration,
the k x k algebra is replicated.
"""
from __future__ import annotations

import time
from typing import Optional, Tuple

import torch

from .dist import ProbeSharder


class JacobianOperator:
    """J and J^T products on the HIP engine for a fixed (x, t, mask)."""

    def __init__(self, engine, x: torch.Tensor, t, at: float, mask: Optional[torch.Tensor], noise: bool = False):
        self.engine = engine
        self.n = engine.n
        self.masked = mask is not None
        engine.pmp_primal(x.contiguous(), float(t), at, mask, use_et=noise)

    def check_mask(self):
        """The gather list is built on the device and its length L is read back lazily (no host sync at the start
        of a solve): the empty-mask error is raised when L is first needed, at the end of the solve."""
        if self.masked and self.engine.mask_count() == 0:
            # the reference would run the power iteration on a 0-row Jacobian and return NaN directions
            raise ValueError("empty mask: J = d x0_hat[mask] / d x_t has no rows")

    def jvp(self, V: torch.Tensor) -> torch.Tensor:   # [k,n] -> dense masked [k,n]
        return self.engine.pmp_jvp(V)

    def vjp(self, U: torch.Tensor) -> torch.Tensor:   # dense [k,n] -> [k,n]
        return self.engine.pmp_vjp(U)

    def gather(self, U: torch.Tensor) -> torch.Tensor:  # dense [k,n] -> [k,L]
        return self.engine.mask_gather(U)


def subspace_iteration(op, algebra, V0: torch.Tensor, min_iter: int = 10, max_iter: int = 100,
                       convergence_threshold: float = 1e-3, sharder: Optional[ProbeSharder] = None,
                       verbose: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]:
    """Core loop of edit.py:2443-2494 on orthonormal rows ``V0`` [k, n].

    ``op`` provides jvp/vjp/gather, ``algebra`` provides orthonormalize_/convergence
    (the HIP engine in production; tests substitute CPU doubles to exercise the
    sharding logic under gloo).  Returns (U_dense [k,n], s [k], V [k,n], n_iter)
    where U = J V_prev of the last iteration (edit.py:2457 -- the reference
    returns u one iteration behind vT) and s are the singular values of A.
    """
    sharder = sharder or ProbeSharder(None)
    k = V0.shape[0]
    V = V0
    n_done = 0
    U = None
    s = None
    for i in range(max_iter):
        V_prev = V
        lo, hi = sharder.rows(k)
        if hi > lo:
            U_loc = op.jvp(V[lo:hi].contiguous())      # u_i = J v_i            (edit.py:2451-2455)
            A_loc = op.vjp(U_loc)                       # a_i = J^T u_i          (edit.py:2460-2480)
        else:                                           # more ranks than probes: this rank only joins the gather
            U_loc = A_loc = V[0:0].contiguous()
        A = sharder.all_gather_rows(A_loc, k)           # the one collective per iteration
        U = U_loc
        V = A
        s = algebra.orthonormalize_(V)                  # _, s, v = svd(v_)      (edit.py:2482)
        n_done = i + 1
        need_flag = i > min_iter
        if verbose or need_flag:
            dist_close = algebra.convergence(V_prev, V, convergence_threshold).tolist()   # edit.py:2489-2492
            if verbose:
                print(f'power method : {i}-th step convergence : ', dist_close[0])
            if need_flag and dist_close[1] > 0.5:
                if verbose:
                    print('reach convergence threshold : ', dist_close[0])
                break
    U = sharder.all_gather_rows(U, k)
    return U, s, V, n_done


def local_basis(engine, x, t, at, pca_rank: int, mask=None, noise=False, min_iter=10, max_iter=100,
                convergence_threshold=1e-3, v0: Optional[torch.Tensor] = None,
                sharder: Optional[ProbeSharder] = None, verbose=True):
    """Top-``pca_rank`` right singular subspace of J (edit.py:2406-2504).

    ``v0``: optional [n, k] Gaussian matrix standing in for the ``torch.randn``
    draw of edit.py:2435 (parity tests inject it).  Returns (u [L,k], s [k],
    vT [k,n], n_iter) with the reference's conventions: ``s`` is the square
    root of the singular values of A = U^T J (edit.py:2500).
    """
    n = engine.n
    dev = x.device
    time_s = time.time()
    if v0 is None:
        v0 = torch.randn(n, pca_rank, device=dev, dtype=torch.float32)      # edit.py:2435
    V = v0.to(device=dev, dtype=torch.float32).T.contiguous()               # rows = probes
    engine.qr_rows_(V)                                                       # edit.py:2436 (thin QR)
    op = JacobianOperator(engine, x, t, at, mask, noise)
    U, s, V, n_iter = subspace_iteration(op, engine, V, min_iter, max_iter, convergence_threshold,
                                         sharder=sharder, verbose=verbose)
    op.check_mask()
    u = op.gather(U).T.contiguous()                                          # [L, k]  (edit.py:2500-2502)
    if verbose:
        torch.cuda.synchronize()
        print('power method runtime ==', time.time() - time_s)
    return u, s.sqrt(), V, n_iter
