mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics_fused or forward_jvp_vjp or headline or pipeline" > gpurun_out/r3e/pytest1.txt 2>&1
tail -15 gpurun_out/r3e/pytest1.txt
for m in 1 0 1 0; do LOCO_FUSE_STATS=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$m', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity']['cos_min'])"; done | tee gpurun_out/r3e/ab.txt
python -m pytest tests/test_gpu_tloco.py -x -q -m gpu -k "config5" > gpurun_out/r3e/pytest2.txt 2>&1; tail -3 gpurun_out/r3e/pytest2.txt
