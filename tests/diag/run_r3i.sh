R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3i
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flash_attention or statistics_fused" > gpurun_out/r3i/pytest1.txt 2>&1
grep -E "flash J V|rel-L2|passed|failed|^E " gpurun_out/r3i/pytest1.txt | tail -20
for m in 1 0 1 0; do LOCO_FLASH_ATTN=$m python bench.py --workload tloco_if64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-e2e --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flash=$m tloco_if64', d['ms_per_step'], d['singular_values'][:3])"; done | tee gpurun_out/r3i/ab.txt
