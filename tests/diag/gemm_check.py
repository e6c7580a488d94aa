"""Diagnostic (by hand, diag build): the DMA-fed 1x1 GEMM (conv_gemm_kernel.h) against the per-pixel 1x1 kernel on the Stable
Diffusion transformer's layer shapes, through `loco_bench_conv` (random operands, same seeds): one child process per setting of
LOCO_CONV_GEMM, outputs compared bit for bit where neither side splits K, to rounding where one does.
    LOCO_HIP_LIB=.../libloco_hip_diag.so python3 tests/diag/gemm_check.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = ((320, 960, 64), (320, 320, 64), (320, 2560, 64), (1280, 320, 64), (640, 1920, 32), (640, 5120, 32), (2560, 640, 32),
          (1280, 3840, 16), (1280, 10240, 16), (5120, 1280, 16), (2560, 320, 64), (960, 320, 64), (512, 1536, 64), (512, 512, 64))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    from loco_edit_amd.config import CELEBA_DDPM, synth_params
    eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
    eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
    eng.set_precision("bf16x3")
    B = int(sys.argv[3]); MODE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    res = {}
    for cin, cout, hw in SHAPES:
        us = eng.bench_conv(cin, cout, hw, hw, B, MODE, 1, -1, 2) * 1e3
        res[(cin, cout, hw)] = (us, eng.debug_tensor("bench_out", cout * hw * hw * B).cpu())
    torch.save(res, sys.argv[2])
    sys.exit(0)
import torch
B = sys.argv[1] if len(sys.argv) > 1 else "5"
MODE = sys.argv[2] if len(sys.argv) > 2 else "0"      # 0 raw input, 2 GroupNorm affine (attention norm -> q, k, v)
out = {}
import shutil, tempfile
TMP = tempfile.mkdtemp(prefix="gemm_check_")      # private to this run
for v in ("0", "1"):
    f = os.path.join(TMP, f"{v}.pt")
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, B, MODE], check=True, env=dict(os.environ, LOCO_CONV_GEMM=v))
    out[v] = torch.load(f)
shutil.rmtree(TMP, ignore_errors=True)
for k in out["0"]:
    (u0, a), (u1, b) = out["0"][k], out["1"][k]
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    fin = torch.isfinite(a) & torch.isfinite(b)
    d = (a[fin] - b[fin]).abs().max().item() if fin.any() else float("nan")
    ref = a[fin].abs().max().item() if fin.any() else float("nan")
    nbad = int(((a - b).abs() > 1e-3 * ref)[fin].sum())
    print(f"{k[0]:5d}->{k[1]:5d} @{k[2]:2d} B={B} mode {MODE}: per-pixel {u0:7.1f} us, gemm {u1:7.1f} us | bit-identical {same} | max|diff| {d:.3e} "
          f"(|ref|max {ref:.3e}, {int((~fin).sum())} non-finite, {nbad} elements off by > 1e-3 |ref|max)")
