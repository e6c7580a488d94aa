R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for FB in 1 25; do
if [ $FB = 1 ]; then PROG="$R/tests/diag/fwd_b1_time.py 1 50"; else PROG="$R/tests/diag/decode_b25.py"; fi
rocprofv3 --kernel-trace --stats -d $O/fb$FB -o f --output-format csv -- python3 $PROG > $O/fb$FB.log 2>&1
grep "B=" $O/fb$FB.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/fb$FB/f_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
ne = 53 if $FB == 1 else 12
print(f"B=$FB kernel time total {tot/1e6:.1f} ms over {n} launches => {tot/1e6/ne:.3f} ms per evaluation, {n/ne:.0f} launches per evaluation")
for r in rows[:22]: print('  ', r['Name'][:95], r['Calls'], f"{float(r['TotalDurationNs'])/1e6/ne:.3f} ms/eval", f"{float(r['AverageNs'])/1e3:.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
done
