R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_quick.json 2> $O/bench_quick.err
tail -c 1500 $O/bench_quick.json
bash tests/diag/pmc_conv_r04.sh r04a > $O/pmc.log 2>&1
tail -40 $O/pmc.log
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=25 > $O/pytest_gpu.txt 2>&1
tail -45 $O/pytest_gpu.txt
