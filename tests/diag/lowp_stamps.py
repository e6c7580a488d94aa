"""Diagnostic (by hand; needs tests/diag/lib/libloco_hip_stamp.so = the diag build with conv_bf16_inst_c.hip / _i.hip compiled
-DLOCO_DUAL_STAMP): where a 128 x 256 tile of the lock-step kernel spends its cycles.  Phase stamps (s_memtime, wave 0) of every
workgroup of one launch of the Cin -> 128 tangent conv at 256^2, 5 probes: 0 start | 1 index setup done | 2 prologue done |
3 stage loop done | 4 epilogue done.  LOCO_DUAL_WHATIF bits: 2 halo loads collapsed, 4 weight DMAs collapsed, 8 no conversions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
os.environ["LOCO_CONV_DUAL"] = "0"
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision("bf16x3")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = 5
us = eng.bench_conv(cin, 128, 256, 256, B, mode, 9, 5, 3) * 1e3
n = 256 * B
raw = eng.debug_tensor("workspace", n * 8 * 2).view(torch.int64).view(n, 8).cpu()
d = raw[:, 1:5] - raw[:, 0:4]
names = ["setup", "prologue", "stage loop", "epilogue"]
print(f"[whatif={os.environ.get('LOCO_DUAL_WHATIF', '0')}] lock-step kernel, mode {mode} cin {cin}: {us:.1f} us per launch; cycles per tile phase "
      f"(median / mean / max over {n} tiles):")
for i, nm in enumerate(names):
    c = d[:, i].double()
    print(f"  {nm:10s} {c.median().item():9.0f} {c.mean().item():9.0f} {c.max().item():9.0f}")
tot = (raw[:, 4] - raw[:, 0]).double()
print(f"  {'tile':10s} {tot.median().item():9.0f} {tot.mean().item():9.0f} {tot.max().item():9.0f}   (matrix work alone: {cin // 16 * 9 * 12 * 2 * 32} cycles)")
span = (raw[:, 4].max() - raw[:, 0].min()).item()
print(f"  first start -> last end: {span} cycles = {span / n * 256:.0f} per tile slot")
