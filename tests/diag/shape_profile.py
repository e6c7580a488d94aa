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
V = torch.randn(5, CELEBA_DDPM.n, generator=torch.Generator().manual_seed(2)).to(dev)
U = eng.pmp_jvp(V); A = eng.pmp_vjp(U); torch.cuda.synchronize()
eng.profile_enable(2)
U = eng.pmp_jvp(V); A = eng.pmp_vjp(U)
rep = eng.profile_report()
eng.profile_enable(False)
tot = sum(v["ms"] for v in rep.values())
print(f"conv total {tot:.2f} ms for one JVP+VJP (k=5)")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 45      # rows; argv[2]: substring filter on the key ("|t1_" = the 1x1 maps)
F = sys.argv[2] if len(sys.argv) > 2 else ""
if F: print(f"rows matching {F!r}: {sum(v['ms'] for k, v in rep.items() if F in k):.2f} ms")
for k, v in [kv for kv in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]) if F in kv[0]][:N]:
    print(f"{k:80s} n={v['launches']:3d} ms={v['ms']:7.3f} ({100*v['ms']/tot:4.1f}%) {v['flops']/v['ms']/1e9:7.1f} TF/s")
