"""Diagnostic (by hand, diag build): the Stable Diffusion transformer's linear layers as the 1x1 conv kernel runs them
(raw input, 5 probes): time and TFLOP/s per shape.  LOCO_HIP_LIB=.../libloco_hip_diag.so python tests/diag/conv1x1_sd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
for cin, cout, hw, what in ((320, 960, 64, "qkv"), (320, 320, 64, "proj / to_out"), (320, 2560, 64, "ff1"), (1280, 320, 64, "ff2"),
                            (640, 1920, 32, "qkv"), (640, 5120, 32, "ff1"), (2560, 640, 32, "ff2"),
                            (1280, 3840, 16, "qkv"), (1280, 10240, 16, "ff1"), (5120, 1280, 16, "ff2")):
    try:
        us = eng.bench_conv(cin, cout, hw, hw, 5, 0, 1, -1, 6) * 1e3
    except Exception as e:
        print(f"1x1 {cin}->{cout} @{hw}: {e}"); continue
    gf = 2.0 * cin * cout * hw * hw * 5 / 1e9
    print(f"1x1 {cin:5d}->{cout:5d} @{hw:2d} ({what}): {us:8.1f} us  {gf / us * 1e3:6.1f} TFLOP/s", flush=True)
