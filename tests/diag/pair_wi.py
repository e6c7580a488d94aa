"""Diagnostic (by hand): what-if builds of the tap-pair conv kernel (tests/diag/libwi/libloco_pk_*.so: -DPKW_NOBAR / _NOCONV / _NODMA /
_NOLOAD, results wrong by construction) against the diagnostics build: time of the dominant shape, random and all-zero operands.
python3 tests/diag/pair_wi.py [iters]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import loco_edit_amd  # noqa
    import loco_edit_amd.hip as H
    from loco_edit_amd.config import CELEBA_DDPM, synth_params
    eng = H.LocoEngine(CELEBA_DDPM, max_batch=8)
    eng.load_state_dict(synth_params(CELEBA_DDPM, 0))
    eng.set_precision("bf16x3")
    it = int(sys.argv[2])
    print(" ".join(f"mode{m}={eng.bench_conv(128, 128, 256, 256, 5, m, 9, 5, it) * 1e3:.1f}us" for m in (0, 1, 3, 4)), flush=True)
    sys.exit(0)
iters = sys.argv[1] if len(sys.argv) > 1 else "300"
libs = [os.path.join(ROOT, "loco-edit_amd", "libloco_hip_diag.so")] + sorted(glob.glob(os.path.join(ROOT, "tests", "diag", "libwi", "*.so")))
for zero in ("0", "3"):
    for pair, lib in [("0", libs[0])] + [("1", l) for l in libs]:
        env = dict(os.environ, LOCO_CONV_PAIR=pair, LOCO_HIP_LIB=lib, LOCO_BENCH_ZERO=zero)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", iters], env=env, capture_output=True, text=True)
        print(f"zero={zero} pair={pair} {os.path.basename(lib):28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
