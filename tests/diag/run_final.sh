R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
python -m pytest tests -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1
tail -3 gpurun_out/final/pytest_gpu.txt
bash profiles/collect.sh r03
