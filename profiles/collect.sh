#!/bin/bash
# Round profile collection (run through gpurun from the repo root):  bash profiles/collect.sh r06
# Everything lands under gpurun_out/<tag>/ (the only directory that travels back); copy the summaries into profiles/.
# 1. default bench (CPU baseline, e2e phases, extra workloads)               -> bench.json
# 2. rocprofv3 --kernel-trace --stats of the headline bench ON ONE STREAM     -> stats/   (+ bench_under_rocprof.json)
#    (the default run puts the probe groups of a pass on two HIP streams since round 6: kernel durations under overlap are not
#    kernel durations; bench.py's own per-kernel profile -- `roofline` -- is taken on one stream too, so the averages agree)
# 3. PMC passes FETCH_SIZE / WRITE_SIZE, separate runs (MI355X guide)        -> pmc_fetch/, pmc_write/
# 4. profiles/make_traffic.py on 3.                                          -> traffic.json (+ per-kernel csv), stamped with
#    the sha256 of the library it was measured on; the script FAILS if that stamp does not match the library in the tree
# 5. the same kernel-trace summary for the other workloads (configs 3, 4, 5) -> stats_<workload>/
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --streams 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-profile --no-e2e --no-extra > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-profile --no-e2e --no-extra > /dev/null 2> $O/pmc_write.err
python3 $R/profiles/make_traffic.py $O/pmc_fetch $O/pmc_write $tag $O > $O/traffic_summary.txt 2>&1 || { echo "make_traffic failed"; cat $O/traffic_summary.txt; exit 1; }
python3 $R/profiles/check_traffic.py $O/traffic.json $R/loco-edit_amd/libloco_hip.so || exit 1
for wl in p2_k64 tloco_if_i_m tloco_sd15; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$wl -o s --output-format csv -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_${wl}_under_rocprof.json 2> $O/stats_$wl.err
done
# keep the merge small: the per-launch traces are large, the summaries are what is judged
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +30M -delete
tail -c 600 $O/bench.json
ls $O
