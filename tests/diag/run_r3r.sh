R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3r
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3r/stats_b25 -o s --output-format csv -- python3 $R/tests/diag/decode_b25.py > $R/gpurun_out/r3r/out.txt 2> $R/gpurun_out/r3r/err.txt
rm -f $R/gpurun_out/r3r/stats_b25/*kernel_trace.csv
cat $R/gpurun_out/r3r/out.txt | tail -2
B=1 python3 $R/tests/diag/decode_b25.py | tail -1
