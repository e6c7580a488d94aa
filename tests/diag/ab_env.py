"""Diagnostic (by hand): whole-step A/B of one environment knob on one box, interleaved.
python tests/diag/ab_env.py NAME v1,v2,... workload [workload ...]"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, vals, WL = sys.argv[1], sys.argv[2].split(","), sys.argv[3:] or ["celeba_top5"]
for wl in WL:
  for rep in range(2):
    for v in vals:
        env = dict(os.environ, **{name: v})
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-e2e", "--no-extra", "--workload", wl], env=env, capture_output=True, text=True)
        try:
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
            print(wl, f"{name}={v}", d["ms_per_step"], d["roofline"]["avg_launch_ms"], (d.get("parity") or {}).get("cos_min"), flush=True)
        except Exception:
            print(wl, v, "FAILED", r.stderr[-300:], flush=True)
