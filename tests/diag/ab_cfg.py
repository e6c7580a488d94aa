"""Diagnostic (by hand): whole-step A/B of several environment configurations on one box, interleaved, two repetitions.
python tests/diag/ab_cfg.py "A=1,B=0" "A=0" ... [-- workload ...]"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
WL = ["celeba_top5"]
if "--" in args:
    WL = args[args.index("--") + 1:]
    args = args[:args.index("--")]
for wl in WL:
  for rep in range(2):
    for cfg in args:
        env = dict(os.environ, **dict(kv.split("=") for kv in cfg.split(",") if kv))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-e2e", "--no-extra", "--workload", wl], env=env, capture_output=True, text=True)
        try:
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
            ro = d["roofline"]
            print(wl, cfg, d["ms_per_step"], ro["kernel"], ro["avg_launch_ms"], (d.get("parity") or {}).get("cos_min"),
                  d["clock"]["sclk_mhz_avg_over_timed_region"], flush=True)
        except Exception:
            print(wl, cfg, "FAILED", r.stderr[-300:], flush=True)
