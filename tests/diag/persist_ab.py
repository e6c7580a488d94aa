"""Diagnostic (by hand, GPU box): persistent-workgroup form of the 8-wave conv launches (LOCO_CONV_PERSIST=N)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "tests/diag/lib")
code = ("import os,sys; sys.path.insert(0, %r); import loco_edit_amd; from loco_edit_amd.config import CELEBA_DDPM, synth_params; "
        "from loco_edit_amd.hip import LocoEngine; e = LocoEngine(CELEBA_DDPM, max_batch=8); e.load_state_dict(synth_params(CELEBA_DDPM, 0));\n"
        "for p in ('bf16x3', 'f16'):\n"
        "    e.set_precision(p)\n"
        "    for (ci, co) in ((128, 128), (256, 128)):\n"
        "        ms = e.bench_conv(ci, co, 256, 256, 5, 3, 9, -1, 20)\n"
        "        print(os.environ.get('TAG'), p, ci, co, f'{ms*1e3:.1f} us', flush=True)\n") % (ROOT,)
cfgs = [("base", "libloco_rev_base.so", "0"), ("loop-only", "libloco_rev_persist.so", "0"), ("persist256", "libloco_rev_persist.so", "256"),
        ("persist512", "libloco_rev_persist.so", "512")]
for tag, lib, n in cfgs:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCO_HIP_LIB=os.path.join(LIB, lib), LOCO_CONV_PERSIST=n, TAG=tag))
for rep in range(2):
    for tag, lib, n in cfgs[:3]:
        env = dict(os.environ, LOCO_HIP_LIB=os.path.join(LIB, lib), LOCO_CONV_PERSIST=n)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-e2e", "--no-extra"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
            print(tag, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["parity"]["cos_min"], flush=True)
        except Exception:
            print(tag, "FAILED", r.stderr[-300:], flush=True)
