R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3k
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_attention" > gpurun_out/r3k/pytest1.txt 2>&1
grep -E "passed|failed|^E " gpurun_out/r3k/pytest1.txt | tail -5
for wl in tloco_if64 tloco_sd; do python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-e2e --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['singular_values'][:3])"; done | tee gpurun_out/r3k/ab.txt
