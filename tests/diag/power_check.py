"""Diagnostic (by hand, diag build): is the 3x3 conv's wall time set by the chip's clock management?  The same launch on random and
on all-zero operands (LOCO_BENCH_ZERO: bit 1 activations, bit 2 weights), >= 0.3 s of back-to-back launches per number, one
process per setting.  python tests/diag/power_check.py [prec] [mode] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision(prec)
for rep in range(3):
    us = eng.bench_conv(128, 128, 256, 256, 5, mode, 9, 5, iters) * 1e3
    print(f"[zero={os.environ.get('LOCO_BENCH_ZERO', '0')}] {prec} 128->128 @256 mode {mode} x{iters}: {us:.1f} us", flush=True)
