R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3u
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tr -o s --output-format csv -- python3 $R/bench.py --workload tloco_sd15 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra > $O/bench.json 2> $O/err.txt
python3 $R/tests/diag/trace_groups.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 80 > $O/groups.txt
find $O -name "*kernel_trace.csv" -delete
head -90 $O/groups.txt
