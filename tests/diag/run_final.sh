R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
python -m pytest tests -q -m gpu --durations=15 > gpurun_out/final/pytest_gpu.txt 2>&1
tail -22 gpurun_out/final/pytest_gpu.txt
bash profiles/collect.sh r04
