R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-tloco_sd15}
O=$R/gpurun_out/r3u_$WL
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tr -o s --output-format csv -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra > $O/bench.json 2> $O/err.txt
python3 $R/tests/diag/trace_groups.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 90 > $O/groups.txt
find $O -name "*kernel_trace.csv" -delete
grep -v conv_mfma $O/groups.txt | head -50
