R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3f
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics_fused" > gpurun_out/r3f/pytest1.txt 2>&1
grep -E "rel-L2|passed|failed" gpurun_out/r3f/pytest1.txt | tail -8
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export LOCO_FUSE_STATS=$m
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3f/stats_fuse$m -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3f/bench_fuse$m.json 2> $R/gpurun_out/r3f/err_fuse$m.txt
done
ls $R/gpurun_out/r3f/stats_fuse1
