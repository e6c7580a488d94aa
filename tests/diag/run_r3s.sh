R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3s
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics_fused or headline or forward_jvp" > gpurun_out/r3s/pytest1.txt 2>&1
tail -3 gpurun_out/r3s/pytest1.txt
bash profiles/collect.sh r03
