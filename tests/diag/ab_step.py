"""Diagnostic (by hand): whole-step A/B of every library under tests/diag/lib on one box (LOCO_HIP_LIB), interleaved."""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WL = sys.argv[1:] or ["celeba_top5"]      # python tests/diag/ab_step.py [workload ...]
libs = sorted(f for f in os.listdir(os.path.join(ROOT, "tests/diag/lib")) if f.endswith(".so"))
for wl in WL:
  for rep in range(2):
    for l in libs:
        env = dict(os.environ, LOCO_HIP_LIB=os.path.join(ROOT, "tests/diag/lib", l))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-e2e", "--no-extra", "--workload", wl], env=env, capture_output=True, text=True)
        try:
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
            print(wl, l, d["ms_per_step"], d["roofline"]["avg_launch_ms"], (d.get("parity") or {}).get("cos_min"), flush=True)
        except Exception:
            print(l, "FAILED", r.stderr[-300:], flush=True)
