R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3g
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics_fused or forward_jvp_vjp or headline or pipeline or eta1" > gpurun_out/r3g/pytest1.txt 2>&1
grep -E "rel-L2|passed|failed" gpurun_out/r3g/pytest1.txt | tail -8
for m in 1 0 1 0; do LOCO_FUSE_STATS=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['e2e']; print('fuse=$m', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['parity']['cos_min'], e['inversion_s'], e['to_t_s'], e['two_solves_s'], e['decode_all_directions_s'])"; done | tee gpurun_out/r3g/ab.txt
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export LOCO_FUSE_STATS=$m
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3g/stats_fuse$m -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --no-profile > $R/gpurun_out/r3g/bench_fuse$m.json 2> $R/gpurun_out/r3g/err_fuse$m.txt
done
