#!/bin/bash
# IF denoiser: the remaining new tests, the headline A/B against the pre-IF library, the new bench workload
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_if.py tests/test_gpu_tloco.py -m gpu -x -q -k "cli_shipped_if or full_width" --durations=5 2>&1 | tail -25 > gpurun_out/if2_tests.txt
cat gpurun_out/if2_tests.txt
timeout 900 python3 tests/diag/ab_step.py celeba_top5 2>&1 | tee gpurun_out/if2_ab.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --workload tloco_if_i_m 2>&1 | tail -3 | tee gpurun_out/if2_bench.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --workload tloco_if64 2>&1 | tail -3 | tee -a gpurun_out/if2_bench.txt
