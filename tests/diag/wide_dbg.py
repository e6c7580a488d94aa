import os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, loco_edit_amd, loco_oracle as orc
from loco_edit_amd.config import WIDE_LDM, TINY_LDM, synth_params
from loco_edit_amd.hip import LocoEngine
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).flatten(1).norm(dim=1) / b.flatten(1).norm(dim=1)).tolist()
cfg = WIDE_LDM
params = synth_params(cfg, 0); p = orc.to_torch(params)
eng = LocoEngine(cfg, max_batch=4, device=torch.device("cuda:0")); eng.load_state_dict(params); eng.set_precision("f32")
g = torch.Generator().manual_seed(43)
z = torch.randn(1, 4, cfg.resolution, cfg.resolution, generator=g)
ctx = torch.randn(cfg.context_len, cfg.context_dim, generator=g)
t = torch.tensor(603.0)
eng.set_context(ctx.cuda().contiguous())
f = lambda z_: orc.unet_forward_adm(p, cfg, z_, t, context=ctx)
for name, zb in (("z", z), ("cat3", torch.cat([z, 0.5 * z.flip(-1), z + 0.2], dim=0)), ("z2", torch.cat([z, z])),
                 ("flip", 0.5 * z.flip(-1)), ("plus", z + 0.2), ("randn3", torch.randn(3, 4, 8, 8, generator=g))):
    with torch.no_grad():
        ref = f(zb)
    out = eng.unet_forward(zb.cuda().contiguous(), float(t))
    print(name, ["%.1e" % r for r in rel(out, ref)], float(ref.norm()), float(out.norm()), flush=True)
