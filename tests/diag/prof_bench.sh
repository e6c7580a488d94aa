cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_now -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_now.log 2>&1
tail -n 2 $R/gpurun_out/prof_now.log | cut -c1-300
