"""Diagnostic (by hand): per-phase cycle shares of the bf16x3 conv stage loop (needs libloco_hip_stamp.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
H._LIB_PATH = os.path.join(ROOT, "loco-edit_amd", "libloco_hip_stamp.so")
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision("bf16x3")
names = ["Ybarrier->X", "X work", "X barrier", "Y mfma", "Y vmcnt", "-", "epilogue"]
for mode in (3,):
    for tile in (5,):
        ms = eng.bench_conv(128, 128, 256, 256, 5, mode, 9, tile, 3)
        v = eng.debug_read_scratch(64)
        print(f"mode {mode} tile {tile}: {ms*1e3:.1f} us")
        for w in range(8 if tile == 5 else 4):
            t = v[w * 8:w * 8 + 7]
            tot = sum(t)
            print("   wave", w, " ".join(f"{n}={x/1e3:.1f}k({100*x/tot:.0f}%)" for n, x in zip(names, t)), f"total={tot/1e3:.0f}k")
