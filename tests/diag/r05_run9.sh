# round 5, run 9: per-kernel durations with / without probe-batched statistics (rocprofv3 --kernel-trace --stats of the headline)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run9; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LOCO_CONV_DUAL=0
for PB in 0 1; do
export LOCO_TSTATS_PB=$PB
rocprofv3 --kernel-trace --stats -d $O/stats_pb$PB -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_pb$PB.json 2> $O/stats_pb$PB.err
find $O -name "*kernel_trace.csv" -delete
f=$(find $O/stats_pb$PB -name "*kernel_stats.csv" | head -1)
echo "== PB=$PB"; grep -E "gn_|splitk" $f | cut -d, -f1-5 | sed 's/(float const.*)"/"/' | cut -c1-120
done
