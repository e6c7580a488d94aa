"""Diagnostic (by hand, GPU box): tloco_sd step with the attention products on the exact f32 kernel vs the split-bf16 one."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(2):
    for m in ("0", "1"):
        env = dict(os.environ, LOCO_GEMM_LOWP=m)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                            "--no-profile", "--workload", "tloco_sd"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][0])
            print("LOCO_GEMM_LOWP=" + m, d["ms_per_step"], d["singular_values"], flush=True)
        except Exception:
            print(m, "FAILED", r.stderr[-300:], flush=True)
