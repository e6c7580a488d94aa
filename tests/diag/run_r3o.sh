R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3o/stats_sd15 -o s --output-format csv -- python3 $R/bench.py --workload tloco_sd15 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3o/bench_sd15.json 2> $R/gpurun_out/r3o/err.txt
rm -f $R/gpurun_out/r3o/stats_sd15/*kernel_trace.csv
head -20 $R/gpurun_out/r3o/stats_sd15/s_kernel_stats.csv | cut -c1-120
