"""Diagnostic (by hand, GPU box): ms per denoiser evaluation of the DDIM chain, eager launches vs HIP-graph replay.
    python tests/diag/graph_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa: F401
from loco_edit_amd.config import CELEBA_DDPM, FFHQ_P2, synth_params
from loco_edit_amd.hip import LocoEngine

for name, cfg in (("celeba_ddpm", CELEBA_DDPM), ("ffhq_p2", FFHQ_P2)):
    for mode in ("0", "1"):
        os.environ["LOCO_GRAPH"] = mode
        eng = LocoEngine(cfg, max_batch=8)
        eng.load_state_dict(synth_params(cfg, 0))
        for B in (1, 5):
            x = torch.randn(B, 3, 256, 256, device="cuda")
            for _ in range(3):
                x = eng.ddim_step(x, 500.0, 0.05, 0.06)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 40
            for i in range(n):
                x = eng.ddim_step(x, 500.0 - i, 0.05, 0.06)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"{name} graph={mode} B={B}: {dt*1e3:.3f} ms per evaluation", flush=True)
        del eng
