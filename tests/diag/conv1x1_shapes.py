"""Diagnostic (by hand, diag build): per-shape time of the 1x1 conv kernel (raw input) at the headline's shortcut shapes,
5 probes; bytes = fp32 input + output per pixel.  LOCO_HIP_LIB=.../libloco_hip_diag.so python tests/diag/conv1x1_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
for cin, cout, hw in ((256, 128, 256), (128, 256, 256), (256, 128, 128), (384, 256, 64), (512, 256, 64), (512, 256, 32), (768, 512, 32),
                      (1024, 512, 16), (1024, 512, 8), (512, 512, 16), (512, 1536, 16)):
    for B in (5,):
        us = eng.bench_conv(cin, cout, hw, hw, B, 0, 1, -1, 8) * 1e3
        gf = 2.0 * cin * cout * hw * hw * B / 1e9
        mb = 4.0 * (cin + cout) * hw * hw * B / 1e6
        print(f"1x1 {cin}->{cout} @{hw} B={B}: {us:7.1f} us  {gf / us * 1e3:6.1f} TFLOP/s  {mb / us:6.2f} TB/s (in+out)", flush=True)
