"""Diagnostic (by hand, diag build): the ResBlock shortcut's cotangent operator -- a 1x1 conv whose epilogue adds the norm-cotangent
term (ConvArgs::cot_d, LOCO_BENCH_COT=1) -- against the same launch without the term, per shape, with the bytes each form moves;
for every diagnostics library given (default: the in-tree one + tests/diag/libwi/*.so), outputs compared with the first one's.
One process per setting (the switches are read once).    python3 tests/diag/nin_check.py [iters] [lib ...]"""
import glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(128, 256, 256, 5), (128, 256, 128, 5), (256, 512, 64, 5), (128, 384, 128, 5), (256, 128, 128, 4), (512, 512, 32, 5)]
if os.environ.get("NIN_SHAPES"):      # "cin,cout,hw,B;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["NIN_SHAPES"].split(";") if t]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    from loco_edit_amd.config import CELEBA_DDPM, synth_params
    eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
    eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
    eng.set_precision("bf16x3")
    iters = int(sys.argv[3])
    res = {}
    for ci, co, hw, b in SHAPES:
        us = min(eng.bench_conv(ci, co, hw, hw, b, 0, 1, -1, iters) for _ in range(3)) * 1e3
        res[(ci, co, hw, b)] = (us, eng.debug_tensor("bench_out", co * hw * hw * b).cpu())
    torch.save(res, sys.argv[2])
    sys.exit(0)
import torch
iters = sys.argv[1] if len(sys.argv) > 1 else "200"
libs = sys.argv[2:] or [os.path.join(ROOT, "loco-edit_amd", "libloco_hip_diag.so")] + sorted(glob.glob(os.path.join(ROOT, "tests", "diag", "libwi", "*.so")))
TMP = tempfile.mkdtemp(prefix="nin_check_")
for cot in os.environ.get("NIN_COT", "0,1").split(","):
    ref = None
    for lib in libs:
        f = os.path.join(TMP, "o.pt")
        env = dict(os.environ, LOCO_HIP_LIB=os.path.abspath(lib), LOCO_BENCH_COT=cot)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, iters], env=env, check=True)
        res = torch.load(f)
        ref = ref or res
        for (ci, co, hw, b), (us, o) in res.items():
            mb = 4e-6 * hw * hw * (b * ci + b * co * (2 if cot == "1" else 1) + (2 * co if cot == "1" else 0))   # in + out (+ d + the shared {S, xhat} records)
            a = ref[(ci, co, hw, b)][1]
            rel = ((a - o).norm() / a.norm()).item()
            print(f"{os.path.basename(lib):28s} cot={cot} {ci:4d} -> {co:4d} @{hw:3d} B={b}: {us:7.1f} us  {mb:5.0f} MB = {mb / us:5.2f} TB/s | "
                  f"vs first library: {100 * (us / ref[(ci, co, hw, b)][0] - 1):+5.1f} %, rel-L2 {rel:.1e} finite {bool(torch.isfinite(o).all())}", flush=True)
shutil.rmtree(TMP, ignore_errors=True)
