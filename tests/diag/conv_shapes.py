"""Diagnostic (by hand, diag build): per-shape time of the 3x3 conv kernel at 256^2 / 128^2, 5 probes, for the modes of one
pass (0 raw, 1 GroupNorm+SiLU forward, 3 tangent, 4 cotangent) -- one process per environment configuration.
python tests/diag/conv_shapes.py [prec] [modes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 3, 4]
eng.set_precision(prec)
tag = " ".join(f"{k}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("LOCO_CONV") or k.startswith("LOCO_SPEC"))
for cin, cout, hw in ((128, 128, 256), (256, 128, 256), (128, 256, 256), (256, 256, 128)):
    for mode in modes:
        us = eng.bench_conv(cin, cout, hw, hw, 5, mode, 9, 5, 4) * 1e3
        gf = 2.0 * 9 * cin * cout * hw * hw * 5 / 1e9
        print(f"[{tag}] {prec} {cin}->{cout} @{hw} mode {mode}: {us:.1f} us  {gf / us:.1f} GFLOP/us", flush=True)
