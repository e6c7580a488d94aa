R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/fb25 -o f --output-format csv -- python3 $R/tests/diag/decode_b25.py > $O/fb25.log 2>&1
grep "B=" $O/fb25.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/fb25/f_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('gn_', 'kcat', 'splitk')): print('  ', r['Name'][:60], r['Calls'], f"{float(r['TotalDurationNs'])/1e6/12:.3f} ms/eval", f"{float(r['AverageNs'])/1e3:.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
