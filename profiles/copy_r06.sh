#!/bin/bash
# copies the summaries of a `bash tests/diag/run_final.sh` collection (gpurun_out/r06, gpurun_out/final) into profiles/
cd "$(dirname "$0")/.."
G=gpurun_out/r06; P=profiles
grep '^{' $G/bench.json | tail -1 > $P/r06_bf16x3_bench.json
cp $(find $G/stats -name "*kernel_stats.csv" | head -1) $P/r06_bf16x3_kernel_stats.csv
grep '^{' $G/bench_under_rocprof.json | tail -1 > $P/r06_bench_under_rocprof.json
for wl in p2_k64 tloco_if_i_m tloco_sd15; do
  cp $(find $G/stats_$wl -name "*kernel_stats.csv" | head -1) $P/r06_${wl}_kernel_stats.csv
  grep '^{' $G/bench_${wl}_under_rocprof.json | tail -1 > $P/r06_bench_${wl}_under_rocprof.json
done
cp $G/r06_pmc_traffic_per_kernel.csv $P/; cp $G/traffic.json $P/traffic.json
cp $G/pmc_final/conv3x3_tan_pmc_mem_bf16x3.csv $P/r06_conv3x3_tan_pmc_mem_bf16x3.csv 2>/dev/null
tail -25 gpurun_out/final/pytest_gpu.txt > $P/r06_pytest_gpu_tail.txt
python3 $P/check_traffic.py $P/traffic.json loco-edit_amd/libloco_hip.so
