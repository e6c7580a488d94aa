"""GPU bring-up diagnostics (run by hand through gpurun, not collected by pytest):
layer-by-layer comparison of the HIP engine with the CPU oracle, then the
tangent / cotangent passes and the solver algebra against the golden vectors.

    python tests/diag/gpu_bringup.py [tiny|mid|full] ...
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import loco_oracle as orc  # noqa: E402
from loco_edit_amd.config import TINY_DDPM, MID_DDPM, CELEBA_DDPM, TINY_ADM, FFHQ_P2, synth_params  # noqa: E402
from loco_edit_amd.hip import LocoEngine  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item(), (a - b).abs().max().item()


def run(tag, cfg, max_batch=8, layerwise=True):
    print(f"==== {tag}: arch {cfg.arch} res {cfg.resolution} ch {cfg.ch} mult {cfg.ch_mult} "
          f"precision {os.environ.get('LOCO_PRECISION', 'bf16x3')}", flush=True)
    params = synth_params(cfg, seed=0)
    eng = LocoEngine(cfg, max_batch=max_batch)
    eng.load_state_dict(params)
    print(f"   workspace {eng.workspace_bytes()/2**30:.2f} GiB, unet flops {eng.unet_flops()/1e9:.2f} GF")
    gpath = os.path.join(GOLD, f"{tag}.pt")
    g = torch.load(gpath) if os.path.exists(gpath) else None
    dev = torch.device("cuda:0")
    if g is not None:
        x, t = g["x"], g["t"]
    else:
        x = torch.randn(1, 3, cfg.resolution, cfg.resolution, generator=torch.Generator().manual_seed(1))
        s = orc.Scheduler(); s.set_timesteps(100); t = s.timesteps[40]
    xd = x.to(dev)
    eps = eng.unet_forward(xd, float(t))
    torch.cuda.synchronize()
    if g is not None and g.get("eps") is not None:
        print("   forward vs golden eps: rel %.3e max %.3e" % rel(eps, g["eps"]))
    elif g is not None:
        print("   forward vs golden eps samples: rel %.3e max %.3e" %
              rel(eps.reshape(-1)[g["eps_sample_idx"].to(dev)], g["eps_sample"]))
    if layerwise:
        p = orc.to_torch(params)
        tr = {}
        with torch.no_grad():
            eo = orc.denoiser(p, cfg, x, t, trace=tr)
        print("   forward vs oracle eps: rel %.3e max %.3e" % rel(eps, eo))
        for name, ref in tr.items():
            got = eng.debug_tensor(name, ref.numel()).reshape(ref.shape)
            r, m = rel(got, ref)
            flag = "" if r < 1e-4 else "   <<<<<<"
            print(f"      {name:28s} rel {r:.3e} max {m:.3e}{flag}")
    # timing of the forward
    for B in (1, min(5, max_batch)):
        xb = xd.repeat(B, 1, 1, 1).contiguous()
        eng.unet_forward(xb, float(t)); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            eng.unet_forward(xb, float(t))
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        print(f"   forward B={B}: {dt*1e3:.2f} ms  ({eng.unet_flops()*B/dt/1e12:.1f} TF/s)")
    if g is None:
        return eng
    # ---- J V and U^T J
    sched = orc.Scheduler()
    at = float(sched.alpha_at(t))
    mask = g["mask"]
    eng.pmp_primal(xd, float(t), at, mask.to(dev))
    if "V" in g:
        V = g["V"].reshape(g["V"].shape[0], -1).contiguous()
    else:
        v0 = torch.randn(cfg.n, g["JV"].shape[0], generator=torch.Generator().manual_seed(g["v0_seed"]))
        V = torch.linalg.qr(v0)[0].T.contiguous()
    k = V.shape[0]
    Ud = eng.pmp_jvp(V.to(dev))
    Ug = eng.mask_gather(Ud)
    print("   JV vs golden: rel %.3e max %.3e" % rel(Ug, g["JV"]))
    Uin = torch.zeros(k, cfg.n)
    Uin[:, mask.reshape(-1)] = g["JV"]
    Ad = eng.pmp_vjp(Uin.to(dev))
    if "UtJ" in g:
        print("   UtJ vs golden: rel %.3e max %.3e" % rel(Ad, g["UtJ"]))
    else:
        P = torch.randn(cfg.n, 64, generator=torch.Generator().manual_seed(g["UtJ_proj_seed"]))
        print("   UtJ projections vs golden: rel %.3e max %.3e" % rel(Ad.cpu() @ P, g["UtJ_proj"]))
    # adjointness <JV, U> = <V, J^T U>
    Ur = torch.randn(k, cfg.n, generator=torch.Generator().manual_seed(3)).to(dev) * mask.reshape(1, -1).to(dev)
    Vr = torch.randn(k, cfg.n, generator=torch.Generator().manual_seed(4)).to(dev)
    lhs = (eng.pmp_jvp(Vr) * Ur).sum(dim=1)
    rhs = (Vr * eng.pmp_vjp(Ur)).sum(dim=1)
    print("   adjoint test: ", ((lhs - rhs).abs() / lhs.abs().clamp_min(1e-9)).max().item())
    # ---- solver algebra
    A = Ad.clone()
    _, s_t, vh_t = torch.linalg.svd(Ad.cpu().double(), full_matrices=False)
    s = eng.orthonormalize_(A)
    cos = (A.cpu().double() * vh_t).sum(dim=1).abs()
    print("   orthonormalize: s rel %.3e, |cos| min %.6f, orth err %.3e" %
          (rel(s, s_t)[0], cos.min().item(), (A @ A.T - torch.eye(k, device=dev)).abs().max().item()))
    Q = torch.randn(k, cfg.n, generator=torch.Generator().manual_seed(5)).to(dev)
    Q0 = Q.clone()
    eng.qr_rows_(Q)
    qt = torch.linalg.qr(Q0.cpu().double().T)[0].T
    print("   qr_rows: orth err %.3e, |cos| vs torch.qr min %.6f" %
          ((Q @ Q.T - torch.eye(k, device=dev)).abs().max().item(), (Q.cpu().double() * qt).sum(dim=1).abs().min().item()))
    torch.cuda.synchronize()
    for kk in (k,):
        Vk = V[:kk].to(dev).contiguous()
        eng.pmp_jvp(Vk); torch.cuda.synchronize()
        t0 = time.time(); U_ = eng.pmp_jvp(Vk); torch.cuda.synchronize(); t1 = time.time()
        A_ = eng.pmp_vjp(U_); torch.cuda.synchronize(); t2 = time.time()
        F = eng.unet_flops() * kk
        print(f"   jvp k={kk}: {(t1-t0)*1e3:.2f} ms ({F/(t1-t0)/1e12:.1f} TF/s)   vjp: {(t2-t1)*1e3:.2f} ms ({F/(t2-t1)/1e12:.1f} TF/s)")
    return eng


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "mid"]
    print(torch.__version__, torch.cuda.get_device_name(0))
    for w in which:
        if w == "tiny":
            run("tiny", TINY_DDPM)
        elif w == "mid":
            run("mid", MID_DDPM)
        elif w == "tiny_adm":
            run("tiny_adm", TINY_ADM)
        elif w == "p2":
            run("p2_256", FFHQ_P2, max_batch=8, layerwise=("--layers" in sys.argv))
        elif w == "full":
            run("celeba256", CELEBA_DDPM, max_batch=8, layerwise=("--layers" in sys.argv))
        torch.cuda.synchronize()
