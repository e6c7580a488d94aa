"""Diagnostic (by hand, under rocprofv3 --kernel-trace --stats): 10 DDIM steps of a 25-frame batch at 256x256 (the decode leg)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
from loco_edit_amd.scheduler import YHCustomScheduler
B = int(os.environ.get("B", "25"))
eng = LocoEngine(CELEBA_DDPM, max_batch=32)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
s = YHCustomScheduler(); s.set_timesteps(100)
x = torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
import time
for i in range(40, 52):
    if i == 42:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    t = s.timesteps[i]
    x = eng.ddim_step(x, float(t), s.alpha_at(t), s.alpha_at(s.timesteps_next[i]), 0.0, None)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step")
