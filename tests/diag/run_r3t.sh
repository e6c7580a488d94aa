R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3t
cd $R
python tests/diag/ab_step.py celeba_top5 > gpurun_out/r3t/ab.txt 2>&1
cat gpurun_out/r3t/ab.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics_fused or headline or forward_jvp or pipeline" > gpurun_out/r3t/pytest1.txt 2>&1
tail -3 gpurun_out/r3t/pytest1.txt
python tests/diag/decode_b25.py | tail -1; B=1 python tests/diag/decode_b25.py | tail -1
