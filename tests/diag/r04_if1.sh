#!/bin/bash
# first GPU run of the DeepFloyd-IF denoiser tests
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_if.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/if1.txt
cat gpurun_out/if1.txt
