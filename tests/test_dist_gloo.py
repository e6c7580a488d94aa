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


class _RectOp:
    """Dense rectangular J [n_out, n] (the latent operator's shape: rows of J V are wider than the probes)."""

    def __init__(self, n, n_out):
        self.J = torch.randn(n_out, n, generator=torch.Generator().manual_seed(3)) / n ** 0.5
        self.n, self.n_out = n, n_out

    def jvp(self, V):
        return V @ self.J.T

    def vjp(self, U):
        return U @ self.J

    def gather(self, U):
        return U


def _run_rect(rank, world, port, k, q):
    sys.path.insert(0, ROOT)
    import loco_edit_amd  # noqa: F401
    from loco_edit_amd.dist import ProbeSharder
    from loco_edit_amd.solver import subspace_iteration
    if world > 1:                     # (the single-process leg runs inside pytest: leave its thread count alone)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    op = _RectOp(48, 112)
    V0 = torch.linalg.qr(torch.randn(48, k, generator=torch.Generator().manual_seed(5)))[0].T.contiguous()
    U, s, V, n = subspace_iteration(op, _CpuAlgebra(), V0, min_iter=3, max_iter=3, sharder=ProbeSharder("world"),
                                    verbose=False)
    if rank == 0:
        q.put((U.tolist(), s.tolist(), V.tolist()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_more_ranks_than_probes_rectangular_operator():
    """with world > k the probe-less ranks contribute EMPTY blocks to both gathers; the block of J V must have
    the operator's output width (n_out != n for the latent operator), or the all-gather shapes disagree."""
    ctx = mp.get_context("spawn")
    q1 = ctx.Queue()
    _run_rect(0, 1, 0, 1, q1)
    U1, s1, V1 = (torch.tensor(v) for v in q1.get())
    q2 = ctx.Queue()
    procs = [ctx.Process(target=_run_rect, args=(r, 2, 29577, 1, q2)) for r in range(2)]
    for p in procs:
        p.start()
    U2, s2, V2 = (torch.tensor(v) for v in q2.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert U2.shape == (1, 112) and V2.shape == (1, 48)
    assert torch.allclose(U1, U2, rtol=1e-5, atol=1e-7) and torch.allclose(V1, V2, rtol=1e-5, atol=1e-7)
    assert torch.allclose(s1, s2, rtol=1e-5)


# ---------------------------------------------------------------------------
# the CLI under torchrun: process group from the environment, rank 0's seed everywhere, rank 0 alone
# writes, file-existence branches agreed, uneven probe shards
def _run_cli(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCO_DIST_BACKEND="gloo")
    os.chdir(tmp)
    import loco_edit_amd  # noqa: F401
    import loco_edit_amd.edit as ledit
    from loco_edit_amd import main as lmain
    from loco_edit_amd.dist import ProbeSharder

    class StubEdit(ledit.EditUncondDiffusion):
        def __init__(self, args):
            self.args, self.sharder = args, ProbeSharder("world")
            self.result_folder = args.result_folder

        def run_edit_null_space_projection(self, idx, **kw):
            p = os.path.join(self.result_folder, "vT-test.pt")
            e1 = self._exists(p)
            self._save(torch.full((1, 4), float(self.sharder.rank)), p)
            self.sharder.barrier()
            e2 = self._exists(p)
            v = self._load(p)
            # 5 probes on 2 ranks: shards of 3 and 2 rows, gathered in probe order
            lo, hi = self.sharder.rows(5)
            rows = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
            g = self.sharder.all_gather_rows(rows, 5)
            return dict(rank=self.sharder.rank, seed=self.args.seed, device=str(self.args.device), e1=e1, e2=e2,
                        loaded=float(v[0, 0]), rows=(lo, hi), gathered=g[:, 0].tolist(),
                        draw=float(torch.rand(())))

    ledit.EditUncondDiffusion = StubEdit
    out = lmain.main(["--device", "cpu", "--seed", "0", "--performance_boosting_t", "0.2", "--result_folder", tmp,
                      "--run_edit_null_space_projection", "True", "--pca_rank", "5"])
    q.put(out)


def test_cli_two_ranks_gloo(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_cli, args=(r, 2, 29583, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120), q.get(timeout=120)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    assert a["seed"] == b["seed"] != 0 and a["draw"] == b["draw"]          # rank 0's drawn seed on both ranks
    assert (a["e1"], b["e1"], a["e2"], b["e2"]) == (False, False, True, True)
    assert a["loaded"] == b["loaded"] == 0.0                                 # only rank 0 wrote; both hold its tensor
    assert a["rows"] == (0, 3) and b["rows"] == (3, 5)
    assert a["gathered"] == b["gathered"] == [0.0, 1.0, 2.0, 3.0, 4.0]
    pts = [f for _, _, fs in os.walk(tmp_path) for f in fs if f.endswith(".pt")]
    assert pts == ["vT-test.pt"]
