#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_if.py -m gpu -x -q --durations=5 2>&1 | tail -25 > gpurun_out/if3_tests.txt
cat gpurun_out/if3_tests.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --workload tloco_if_i_m 2>&1 | tail -1 | cut -c1-900 | tee gpurun_out/if3_bench.txt
LOCO_FLASH_ATTN=0 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --workload tloco_if_i_m 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/if3_bench.txt
