R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
python -m pytest tests -q -m gpu --durations=15 > gpurun_out/final/pytest_gpu.txt 2>&1
tail -22 gpurun_out/final/pytest_gpu.txt
bash profiles/collect.sh r06
# memory-side + SQ counters of the dominant conv shape on the final library (needs the diag build in the snapshot)
python3 tests/diag/pmc_conv_mem.py r06/pmc_final bf16x3 3 5 conv_
