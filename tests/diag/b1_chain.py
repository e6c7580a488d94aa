"""Diagnostic (by hand): a short B=1 DDIM chain for a kernel trace (rocprofv3 --kernel-trace --output-format csv)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa: F401
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
eng = LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
x = torch.randn(1, 3, 256, 256, device="cuda")
for i in range(12):
    x = eng.ddim_step(x, 500.0 - i, 0.05, 0.06)
torch.cuda.synchronize()
