cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tests/diag/fwd_b1_time.py ${FB:-1} 50 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fb1 -o f --output-format csv -- python3 $R/tests/diag/fwd_b1_time.py ${FB:-1} 50 > $R/gpurun_out/fb1.log 2>&1
grep "B=" $R/gpurun_out/fb1.log
python3 - <<'PY'
import csv,os
R=os.environ['GRAFT_REPO_ROOT']
rows=list(csv.DictReader(open(R+'/gpurun_out/fb1/f_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
print(f"kernel time total {tot/1e6:.1f} ms over {n} launches => {tot/1e6/53:.3f} ms per evaluation, {n/53:.0f} launches per evaluation")
for r in rows[:22]: print(r['Name'][:70], r['Calls'], f"{float(r['TotalDurationNs'])/1e6/53:.3f} ms/eval", f"{float(r['AverageNs'])/1e3:.1f} us")
PY
