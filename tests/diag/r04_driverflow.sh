#!/bin/bash
# what the driver runs at round end: smoke, then the default bench line (timed)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python3 __graft_entry__.py smoke 2>&1 | tail -2
SECONDS=0
python3 bench.py > gpurun_out/driverflow_bench.json 2> gpurun_out/driverflow_bench.err
echo "bench rc=$? wall=${SECONDS}s"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/driverflow_bench.json') if l.startswith('{')][-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
print({k: v.get('ms_per_step') for k, v in d['extra_workloads'].items()})
PY
