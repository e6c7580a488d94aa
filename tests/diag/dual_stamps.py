"""Diagnostic (by hand; needs tests/diag/lib/libloco_hip_stamp.so = the diag build with conv_bf16_inst_i.hip compiled
-DLOCO_DUAL_STAMP): where a dual-tile unit spends its cycles.  Phase stamps (s_memtime, wave 0) of every unit of one launch of the
128 -> 128 tangent conv at 256^2, 4 probes:  0 unit start | 1 index setup done | 2 prologue done (first barrier) | 3 chunk loop
done | 4 drained | 5 epilogue done."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import loco_edit_amd  # noqa
import loco_edit_amd.hip as H
from loco_edit_amd.config import CELEBA_DDPM, synth_params
eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
eng.set_precision("bf16x3")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
us = eng.bench_conv(cin, 128, 256, 256, 4, mode, 9, 5, 3) * 1e3
raw = eng.debug_tensor("workspace", 512 * 8 * 2).view(torch.int64).view(512, 8).cpu()
d = raw[:, 1:6] - raw[:, 0:5]
names = ["setup", "prologue", "chunk loop", "drain", "epilogue"]
print(f"[whatif={os.environ.get('LOCO_DUAL_WHATIF', '0')}] mode {mode} cin {cin}: {us:.1f} us per launch; cycles per unit phase (median / mean / max over 512 units):")
for i, n in enumerate(names):
    c = d[:, i].double()
    print(f"  {n:10s} {c.median().item():9.0f} {c.mean().item():9.0f} {c.max().item():9.0f}")
tot = (raw[:, 5] - raw[:, 0]).double()
print(f"  {'unit':10s} {tot.median().item():9.0f} {tot.mean().item():9.0f} {tot.max().item():9.0f}   (matrix work alone: {cin // 16 * 9 * 24 * 2 * 32} cycles)")
first = raw[:256]; second = raw[256:]
gap = (second[:, 0] - first[:, 5]).double()
print(f"  gap between a workgroup's two units: median {gap.median().item():.0f}")
span = (raw[:, 5].max() - raw[:, 0].min()).item()
print(f"  first start -> last end: {span} cycles")
