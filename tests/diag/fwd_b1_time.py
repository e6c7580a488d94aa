"""Measurement aid (by hand, GPU): wall time per denoiser evaluation at B=1 / B=5 (DDIM loops) for comparison with
the summed kernel time of a rocprofv3 --kernel-trace --stats run of this script."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
eng = LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
x = torch.randn(B, 3, 256, 256, device="cuda:0")
for _ in range(3):
    eng.unet_forward(x, 595.36)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    eng.unet_forward(x, 595.36)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"B={B}: {dt*1e3:.3f} ms per evaluation wall ({n} evaluations)")
