"""CPU, world_size 2 over gloo: the probe-sharded subspace iteration (one
all-gather of the A shards per iteration) equals the single-process run.  The
J / J^T products come from the CPU oracle here (test double for the HIP engine);
what is under test is the sharding / collective logic of the product code."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleOp:
    def __init__(self, oed, x, t, mask):
        import loco_oracle as orc
        self.orc, self.oed, self.x, self.t, self.mask = orc, oed, x, t, mask
        self.n = x.numel()

    def jvp(self, V):
        U = self.orc.jvp_x0(self.oed, self.x, self.t, V.reshape(V.shape[0], *self.x.shape[1:]), mask=self.mask)
        out = torch.zeros(V.shape[0], self.n)
        out[:, self.mask.reshape(-1)] = U
        return out

    def vjp(self, U):
        return self.orc.vjp_x0(self.oed, self.x, self.t, U[:, self.mask.reshape(-1)], mask=self.mask)

    def gather(self, U):
        return U[:, self.mask.reshape(-1)]


class _CpuAlgebra:
    def orthonormalize_(self, A):
        _, s, vh = torch.linalg.svd(A, full_matrices=False)
        idx = vh.abs().argmax(dim=1)
        sign = torch.sign(vh[torch.arange(vh.shape[0]), idx])
        A.copy_(vh * sign[:, None])
        return s

    def convergence(self, a, b, atol):
        return torch.tensor([torch.dist(a, b).item(), float(torch.allclose(a, b, atol=atol))])


def _run(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import loco_edit_amd  # noqa: F401
    import loco_oracle as orc
    from loco_edit_amd.config import TINY_DDPM, synth_params
    from loco_edit_amd.dist import ProbeSharder
    from loco_edit_amd.solver import subspace_iteration
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = TINY_DDPM
    oed = orc.OracleEdit(orc.to_torch(synth_params(cfg, 0)), cfg)
    g = torch.load(os.path.join(ROOT, "tests", "golden", "tiny.pt"))
    op = _OracleOp(oed, g["x"], g["t"], g["mask"])
    V0 = torch.linalg.qr(g["v0"][:, :4])[0].T.contiguous()
    sh = ProbeSharder("world")
    U, s, V, n = subspace_iteration(op, _CpuAlgebra(), V0, min_iter=2, max_iter=2, sharder=sh, verbose=False)
    if rank == 0:
        q.put((U.numpy().tolist(), s.numpy().tolist(), V.numpy().tolist(), n, sh.world))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_equals_single():
    ctx = mp.get_context("spawn")
    q1 = ctx.Queue()
    _run(0, 1, 0, q1)
    U1, s1, V1, n1, w1 = q1.get()
    U1, s1, V1 = torch.tensor(U1), torch.tensor(s1), torch.tensor(V1)
    q2 = ctx.Queue()
    port = 29571
    procs = [ctx.Process(target=_run, args=(r, 2, port, q2)) for r in range(2)]
    for p in procs:
        p.start()
    U2, s2, V2, n2, w2 = q2.get(timeout=300)
    U2, s2, V2 = torch.tensor(U2), torch.tensor(s2), torch.tensor(V2)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert w1 == 1 and w2 == 2 and n1 == n2 == 2
    # fixed shard order + identical per-probe arithmetic => same result up to thread-count rounding
    assert torch.allclose(s1, s2, rtol=1e-5)
    assert (V1 * V2).sum(dim=1).abs().min() > 0.99999
    assert torch.allclose(U1, U2, rtol=1e-4, atol=1e-6)
