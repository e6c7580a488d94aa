R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in tloco_if64 tloco_sd tloco_sd15; do
  rm -rf $O/stats_$wl
  rocprofv3 --kernel-trace --stats -d $O/stats_$wl -o s --output-format csv -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_${wl}_under_rocprof.json 2> $O/stats_$wl.err
  python3 -c "import json; d=json.loads(open('$O/bench_${wl}_under_rocprof.json').read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['value'])"
done
find $O -name "*kernel_trace.csv" -delete
