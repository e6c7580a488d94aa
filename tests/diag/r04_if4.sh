#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 tests/diag/ab_step.py celeba_top5 2>&1 | tee gpurun_out/if4_ab.txt
timeout 900 python3 -m pytest tests/test_gpu_if.py tests/test_gpu_parity.py -m gpu -x -q -k "if_i_m_denoiser_at_size or headline_config_12 or adm or p2_256 or statistics_fused" 2>&1 | tail -6 | tee gpurun_out/if4_tests.txt
