"""Diagnostic (by hand, diag build): the 16x16x32 tap-pair conv kernel (conv_pair_kernel.h, LOCO_CONV_PAIR=1) against the 32x32x16
lock-step kernel on single launches through loco_bench_conv: same synthetic operands, outputs compared (summation order differs:
expected relative difference ~1e-6) and times.    python3 tests/diag/pair_check.py [B] [iters]"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = ((128, 128, 256), (256, 128, 256), (128, 256, 128), (256, 256, 64))
MODES = (0, 1, 3, 4)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    from loco_edit_amd.config import CELEBA_DDPM, synth_params
    B, iters = int(sys.argv[3]), int(sys.argv[4])
    eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
    eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
    eng.set_precision("bf16x3")
    res = {}
    for cin, cout, hw in SHAPES:
        for mode in MODES:
            us = eng.bench_conv(cin, cout, hw, hw, B, mode, 9, 5, iters) * 1e3
            res[(cin, cout, hw, mode)] = (us, eng.debug_tensor("bench_out", cout * hw * hw * B).cpu())
    torch.save(res, sys.argv[2])
    sys.exit(0)
import torch
B = sys.argv[1] if len(sys.argv) > 1 else "3"
iters = sys.argv[2] if len(sys.argv) > 2 else "20"
TMP = tempfile.mkdtemp(prefix="pair_check_")
out = {}
for v in ("0", "1"):
    f = os.path.join(TMP, f"{v}.pt")
    env = dict(os.environ, LOCO_CONV_PAIR=v, LOCO_HIP_LIB=os.path.join(ROOT, "loco-edit_amd", "libloco_hip_diag.so"))
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, B, iters], check=True, env=env)
    out[v] = torch.load(f)
shutil.rmtree(TMP, ignore_errors=True)
ok = True
for k in out["0"]:
    (u0, a), (u1, b) = out["0"][k], out["1"][k]
    fin = bool(torch.isfinite(b).all())
    rel = ((a - b).norm() / a.norm()).item()
    mx = ((a - b).abs().max() / a.abs().max()).item()
    good = fin and rel < 2e-6
    ok = ok and good
    print(f"{k[0]:4d}->{k[1]:4d} @{k[2]:3d} mode {k[3]} B={B}: 32x32x16 {u0:7.1f} us, tap-pair 16x16x32 {u1:7.1f} us ({100 * (u1 / u0 - 1):+5.1f} %) | "
          f"rel-L2 {rel:.2e} max {mx:.2e} finite {fin} {'ok' if good else 'MISMATCH'}", flush=True)
print("pair_check:", "PASS" if ok else "FAIL")
