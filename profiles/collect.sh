#!/bin/bash
# Round profile collection (run through gpurun from the repo root):  bash profiles/collect.sh r02
# 1. default bench (CPU baseline, e2e phases, extra workloads) -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the headline bench     -> gpurun_out/<tag>_stats/
# 3. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs)         -> gpurun_out/<tag>_pmc_{fetch,write}/
# Copy the summaries into profiles/ afterwards (profiles/make_traffic.py for step 3).
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o s --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-e2e --no-extra > /dev/null 2> $R/gpurun_out/${tag}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-e2e --no-extra > /dev/null 2> $R/gpurun_out/${tag}_pmc_write.err
tail -c 600 $R/gpurun_out/${tag}_bench.json
ls $R/gpurun_out/${tag}_stats $R/gpurun_out/${tag}_pmc_fetch $R/gpurun_out/${tag}_pmc_write
