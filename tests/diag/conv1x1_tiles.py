import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, ROOT)
import loco_edit_amd, loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8); eng.load_state_dict(synth_params(CELEBA_DDPM, 0)); eng.set_precision("bf16x3")
for cin, cout, hw in ((128, 256, 256), (256, 128, 256), (128, 128, 256), (256, 256, 128), (512, 512, 64)):
    for tile in (5, 0, 1):
        us = eng.bench_conv(cin, cout, hw, hw, 5, 0, 1, tile, 6) * 1e3
        gb = 5 * hw * hw * 4 * (cin + cout) / 1e9
        print(f"1x1 {cin}->{cout} @{hw} tile {tile}: {us:7.1f} us  {gb / us * 1e3:5.2f} TB/s (in + out)", flush=True)
