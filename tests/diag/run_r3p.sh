R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3p
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_latent.py -x -q -m gpu -k "flash_attention or ldm_unet" -s > gpurun_out/r3p/pytest1.txt 2>&1
grep -E "LDM|passed|failed|^E |Error" gpurun_out/r3p/pytest1.txt | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3p/stats_sd15 -o s --output-format csv -- python3 $R/bench.py --workload tloco_sd15 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3p/bench_sd15.json 2> $R/gpurun_out/r3p/err.txt
rm -f $R/gpurun_out/r3p/stats_sd15/*kernel_trace.csv
python3 -c "import json; d=json.loads(open('$R/gpurun_out/r3p/bench_sd15.json').read().strip().splitlines()[-1]); print('tloco_sd15 under rocprof', d['ms_per_step'])"
