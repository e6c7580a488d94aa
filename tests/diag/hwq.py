"""Diagnostic: does the HIP stream -> hardware queue mapping decide whether the CFG branch streams overlap?
python tests/diag/hwq.py <number of dummy streams created (and used once) before the workload>"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
nd = int(sys.argv[1])
dummies = [torch.cuda.Stream(device="cuda:0") for _ in range(nd)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda:0").sum().item()
sys.argv = ["bench.py", "--workload", "tloco_if64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-extra", "--no-profile"]
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    except SystemExit:
        pass
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(f"dummies {nd} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} -> {d['ms_per_step']} ms")
