"""Tuning aid (by hand): time the replicated k x n solver algebra for the multi-GPU probe counts."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
eng = LocoEngine(CELEBA_DDPM, max_batch=1)
n = CELEBA_DDPM.n
dev = torch.device("cuda:0")
for k in (5, 10, 20, 40, 64):
    A0 = torch.randn(k, n, device=dev)
    for name, fn in (("orthonormalize", lambda A: eng.orthonormalize_(A)), ("qr_rows", lambda A: eng.qr_rows_(A)),
                     ("convergence", lambda A: eng.convergence(A, A0, 1e-4))):
        A = A0.clone(); fn(A); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            A = A0.clone(); fn(A)
        torch.cuda.synchronize()
        print(f"k={k:3d} {name:15s} {(time.time()-t0)/5*1e3:8.3f} ms")
