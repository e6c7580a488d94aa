"""Diagnostic (by hand; stamp build, see r05.sh): where a workgroup of the per-pixel 1x1 kernel spends its cycles.  Stamps
(s_memtime, wave 0) of every workgroup of one launch: 0 start | 3 stage loop done (index setup + prologue + loop) | 4 epilogue done.
  python3 tests/diag/lowp_stamps_1x1.py [cin] [cout] [H] [tile]      (5 probes; tile -1 = the engine's choice)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import SD15_UNET, synth_params
os.environ["LOCO_CONV_GEMM"] = "0"
eng = H.LocoEngine(SD15_UNET, max_batch=5)
eng.load_state_dict(synth_params(SD15_UNET, 0))
eng.set_precision("bf16x3")
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 320
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 960
Hh = int(sys.argv[3]) if len(sys.argv) > 3 else 64
tile = int(sys.argv[4]) if len(sys.argv) > 4 else -1
B = 5
us = eng.bench_conv(cin, cout, Hh, Hh, B, 0, 1, tile, 3) * 1e3
px = 128 if (tile == 0 or (tile < 0 and Hh * Hh <= 16384)) else 256
n = (Hh * Hh // px) * ((cout + 127) // 128) * B
raw = eng.debug_tensor("workspace", n * 8 * 2).view(torch.int64).view(n, 8).cpu()
loop = (raw[:, 3] - raw[:, 0]).double()
epi = (raw[:, 4] - raw[:, 3]).double()
tot = (raw[:, 4] - raw[:, 0]).double()
ok = tot > 0
print(f"1x1 {cin} -> {cout} @{Hh}^2, 5 probes, tile {tile}: {us:.1f} us per launch, {n} workgroups ({int(ok.sum())} stamped); cycles per workgroup (median / mean / max):")
for nm, c in (("setup + prologue + stage loop", loop[ok]), ("epilogue", epi[ok]), ("workgroup", tot[ok])):
    print(f"  {nm:30s} {c.median().item():9.0f} {c.mean().item():9.0f} {c.max().item():9.0f}")
print(f"  matrix work alone: {cin // 16 * 12 * 32} cycles per wave; first start -> last end: {(raw[ok][:, 4].max() - raw[ok][:, 0].min()).item()} cycles")
