"""Diagnostic (by hand, diag build): the headline network's 3x3 conv shapes level by level (5 probes), tangent mode by default:
time, TFLOP/s and share of a pass -- of the KERNEL ALONE: `loco_bench_conv` launches it un-split (no split-K, no tail-probe
split), so the levels below 64 x 64 read far worse here than inside the engine; tests/diag/shape_profile.py has the real
per-shape times of a pass.  LOCO_HIP_LIB=.../libloco_hip_diag.so python tests/diag/conv_levels.py [prec] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng.set_precision(prec)
# (cin, cout, hw, how many such convs one pass of the CelebA-HQ DDPM U-Net has: down conv1/conv2, up conv1 (cat input) / conv2)
shapes = [(128, 128, 256, 4 + 3), (256, 128, 256, 3), (128, 128, 128, 4 + 3 + 2), (256, 128, 128, 3),
          (128, 256, 64, 1), (256, 256, 64, 3 + 3 + 2), (512, 256, 64, 2), (384, 256, 64, 1),
          (256, 256, 32, 4 + 3 + 2), (512, 256, 32, 3), (256, 512, 16, 1), (512, 512, 16, 3 + 3 + 2 + 4), (1024, 512, 16, 2), (768, 512, 16, 1),
          (512, 512, 8, 4 + 3 + 4), (1024, 512, 8, 3)]
tot = 0.0
rows = []
for cin, cout, hw, n in shapes:
    us = eng.bench_conv(cin, cout, hw, hw, 5, mode, 9, -1, 6) * 1e3
    gf = 2.0 * 9 * cin * cout * hw * hw * 5 / 1e9
    rows.append((cin, cout, hw, n, us, gf))
    tot += n * us
for cin, cout, hw, n, us, gf in rows:
    print(f"{prec} mode {mode} {cin:4d}->{cout:3d} @{hw:3d} x{n:2d}: {us:7.1f} us {gf / us * 1e3:7.1f} TFLOP/s  {100 * n * us / tot:5.1f} % of the pass's 3x3 time", flush=True)
print(f"sum {tot / 1e3:.2f} ms per pass")
