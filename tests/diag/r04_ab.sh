#!/bin/bash
# same-box step A/B of the libraries under tests/diag/lib + a few parity tests on the in-tree library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 tests/diag/ab_step.py celeba_top5 2>&1 | tee gpurun_out/ab.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline_config_12 or goldens or golden or p2_256 or statistics_fused" 2>&1 | tail -3 | tee gpurun_out/ab_tests.txt
