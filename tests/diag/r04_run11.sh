R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_under_rocprof.json 2> $O/stats.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/stats/s_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 5 solves -> {tot/1e6/5:.1f} ms per solve")
for r in rows[:28]: print(f"{float(r['Percentage']):5.2f}% {r['Calls']:>6} {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
find $O -name "*kernel_trace.csv" -delete
