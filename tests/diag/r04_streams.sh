#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do for st in 1 2; do
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --streams $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('celeba_top5 streams', d['config'].get('streams'), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity']['cos_min'], d['parity']['s_relerr'])"
done; done 2>&1 | tee gpurun_out/streams.txt
for st in 1 2; do
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --workload p2_k64 --streams $st 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('p2_k64 streams', d['config'].get('streams'), d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/streams.txt
