"""Diagnostic (by hand, GPU box): ms per probe-pass of the headline solve as a function of the probes per pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa: F401
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
from loco_edit_amd.scheduler import YHCustomScheduler
from loco_edit_amd import solver
cfg = CELEBA_DDPM
eng = LocoEngine(cfg, max_batch=16)
eng.load_state_dict(synth_params(cfg, 0))
s = YHCustomScheduler(); s.set_timesteps(100)
t = float(s.timesteps[40]); at = s.alpha_at(t)
x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
mask = torch.zeros(3, 256, 256, dtype=torch.bool); mask[:, 110:130, 70:110] = True
mask = mask.cuda()
for k in (5, 8, 10, 16):
    v0 = torch.randn(cfg.n, k, generator=torch.Generator().manual_seed(7)).cuda()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        solver.local_basis(eng, x, t, at, k, mask=mask, min_iter=12, max_iter=12, v0=v0, verbose=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"k={k}: {dt*1e3:.1f} ms per solve, {dt*1e3/k:.2f} ms per probe", flush=True)

# paired modify / null solves (5 + 5 probes per pass) vs the two solves one after the other vs one 10-probe solve
v5 = torch.randn(cfg.n, 5, generator=torch.Generator().manual_seed(7)).cuda()
v10 = torch.randn(cfg.n, 10, generator=torch.Generator().manual_seed(7)).cuda()
def timeit(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
seq = timeit(lambda: (solver.local_basis(eng, x, t, at, 5, mask=mask, min_iter=12, max_iter=12, v0=v5, verbose=False),
                      solver.local_basis(eng, x, t, at, 5, mask=~mask, min_iter=12, max_iter=12, v0=v5, verbose=False)))
pair = timeit(lambda: solver.local_basis_pair(eng, x, t, at, 5, mask, 5, ~mask, min_iter=12, max_iter=12, v0_a=v5, v0_b=v5, verbose=False))
one10 = timeit(lambda: solver.local_basis(eng, x, t, at, 10, mask=mask, min_iter=12, max_iter=12, v0=v10, verbose=False))
print(f"two solves sequential {seq:.1f} ms, paired {pair:.1f} ms, one 10-probe solve {one10:.1f} ms")
