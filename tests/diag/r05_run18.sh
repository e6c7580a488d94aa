# round 5, run 18: per-kernel durations of config 4 with / without the DMA-fed 1x1 GEMM
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run18; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for G in 0 1; do
export LOCO_CONV_GEMM=$G
rocprofv3 --kernel-trace --stats -d $O/stats_g$G -o s --output-format csv -- python3 $R/bench.py --workload tloco_sd15 --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_g$G.json 2> $O/stats_g$G.err
find $O -name "*kernel_trace.csv" -delete
done
python3 - <<'PY'
import csv, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r05_run18"
for g in (0,1):
    rows=list(csv.DictReader(open(f"{O}/stats_g{g}/s_kernel_stats.csv")))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("GEMM",g,"total kernel ms",round(tot/1e6,1))
    for r in rows[:14]:
        print(f"  {r['Name'].split('(')[0][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us tot {float(r['TotalDurationNs'])/1e6:8.1f} ms {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
