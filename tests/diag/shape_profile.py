"""Tuning aid (by hand): per-layer-shape conv timing inside one real solver iteration."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loco_edit_amd  # noqa
from loco_edit_amd.config import CELEBA_DDPM, synth_params
from loco_edit_amd.hip import LocoEngine
from loco_edit_amd.scheduler import YHCustomScheduler
eng = LocoEngine(CELEBA_DDPM, max_batch=8)
eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
s = YHCustomScheduler(); s.set_timesteps(100)
t = float(s.timesteps[40]); at = s.alpha_at(t)
dev = torch.device("cuda:0")
x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
mask = torch.zeros(3, 256, 256, dtype=torch.bool); mask[:, 110:130, 70:110] = True
eng.pmp_primal(x, t, at, mask.to(dev))
K = int(os.environ.get("SHAPE_PROFILE_K", "5"))      # probes per pass (a two-stream pass runs groups of 3 and 2)
FWD = int(os.environ.get("SHAPE_PROFILE_FWD", "0"))   # > 0: one denoiser evaluation of this many frames instead (the DDIM chains: 1, the decode: 21 / 25)
if FWD:
    eng2 = LocoEngine(CELEBA_DDPM, max_batch=32) if FWD > 8 else eng
    if eng2 is not eng: eng2.load_state_dict(synth_params(CELEBA_DDPM, 0))
    xb = torch.randn(FWD, 3, 256, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    eng2.unet_forward(xb, t); torch.cuda.synchronize()
    eng2.profile_enable(2)
    eng2.unet_forward(xb, t)
    rep = eng2.profile_report()
    eng2.profile_enable(False)
    what = f"one denoiser evaluation of {FWD} frames"
else:
    V = torch.randn(K, CELEBA_DDPM.n, generator=torch.Generator().manual_seed(2)).to(dev)
    U = eng.pmp_jvp(V); A = eng.pmp_vjp(U); torch.cuda.synchronize()
    eng.profile_enable(2)
    U = eng.pmp_jvp(V); A = eng.pmp_vjp(U)
    rep = eng.profile_report()
    eng.profile_enable(False)
    what = f"one JVP+VJP (k={K})"
tot = sum(v["ms"] for v in rep.values())
print(f"conv total {tot:.2f} ms for {what}")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 45      # rows; argv[2]: substring filter on the key ("|t1_" = the 1x1 maps)
F = sys.argv[2] if len(sys.argv) > 2 else ""
if F: print(f"rows matching {F!r}: {sum(v['ms'] for k, v in rep.items() if F in k):.2f} ms")
# "excess": time above a simple bound per launch -- max(flops at 350 TFLOP/s, input + output bytes at 4 TB/s) -- times the launches:
# where a shape sits far from both bounds (under-filled chip, unlucky split-K factor, a tile that does not fit the map)
import re
def bound_us(key, v):
    m = re.search(r"t(\d)_m(\d)_ci(\d+)_co(\d+)_h(\d+)_b(\d+)_s(\d+)", key)
    if not m or v["flops"] <= 0: return 0.0
    taps, mode, ci, co, h, b, sp = map(int, m.groups())
    byts = 4.0 * h * h * b * (ci + co)
    return max(v["flops"] / v["launches"] / 350e12, byts / 4e12) * 1e6
rows = []
for k, v in rep.items():
    if F not in k: continue
    bu = bound_us(k, v)
    rows.append((k, v, bu, v["ms"] * 1e3 - bu * v["launches"]))
order = (lambda r: -r[3]) if os.environ.get("SHAPE_PROFILE_SORT", "ms") == "excess" else (lambda r: -r[1]["ms"])
for k, v, bu, ex in sorted(rows, key=order)[:N]:
    print(f"{k:80s} n={v['launches']:3d} ms={v['ms']:7.3f} ({100*v['ms']/tot:4.1f}%) {v['flops']/v['ms']/1e9:7.1f} TF/s  "
          f"bound {bu:6.1f} us/launch  excess {ex:7.1f} us")
