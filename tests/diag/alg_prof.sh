cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/alg -o a --output-format csv -- python3 $R/tests/diag/algebra_bench.py > $R/gpurun_out/alg.log 2>&1
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/alg/a_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:60], r['Calls'], f"avg {float(r['AverageNs'])/1e3:.1f} us", f"max {float(r['MaxNs'])/1e3:.1f} us")
PY
