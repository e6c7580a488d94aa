R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3j
cd $R
for m in 1 0; do LOCO_FLASH_ATTN=$m python bench.py --workload tloco_sd --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-e2e --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flash=$m tloco_sd', d['ms_per_step'], d['singular_values'][:3])"; done | tee gpurun_out/r3j/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3j/stats_if64 -o s --output-format csv -- python3 $R/bench.py --workload tloco_if64 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3j/bench_if64.json 2> $R/gpurun_out/r3j/err_if64.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3j/stats_sd -o s --output-format csv -- python3 $R/bench.py --workload tloco_sd --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3j/bench_sd.json 2> $R/gpurun_out/r3j/err_sd.txt
head -14 $R/gpurun_out/r3j/stats_if64/s_kernel_stats.csv | cut -c1-110
