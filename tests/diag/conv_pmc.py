"""Diagnostic (by hand, under rocprofv3 --pmc): a few launches of the dominant bf16x3 conv shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
eng.set_precision(prec)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ms = eng.bench_conv(128, 128, 256, 256, B, mode, 9, 5, 3)
print(f"{prec} mode {mode} B {B}: {ms*1e3:.1f} us")
