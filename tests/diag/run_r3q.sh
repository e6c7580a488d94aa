R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3q
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r3q/pytest.txt 2>&1
tail -5 gpurun_out/r3q/pytest.txt
