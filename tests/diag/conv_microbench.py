"""Tuning microbenchmark (by hand through gpurun): time the dominant conv shapes per tile variant."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine

eng = LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
shapes = [(128, 128, 256, 5), (256, 128, 256, 5), (128, 128, 128, 5), (256, 256, 64, 5)]
tiles = [int(t) for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["-1"])]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
eng.set_precision(prec)
only_mode = int(os.environ.get("MODE", "-1"))
shapes = shapes[:int(os.environ.get("NSHAPES", "4"))]
for (cin, cout, hw, B) in shapes:
    for mode in (0, 1, 3, 4):
        if only_mode >= 0 and mode != only_mode:
            continue
        for tile in tiles:
            ms = eng.bench_conv(cin, cout, hw, hw, B, mode, 9, tile, int(os.environ.get("ITERS", "20")))
            fl = 2.0 * cin * cout * 9 * hw * hw * B
            print(f"{prec} cin {cin} cout {cout} {hw}x{hw} B{B} mode {mode} tile {tile}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
