R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or full_size_forward_vs_golden or headline_config or statistics_fused or forward_jvp_vjp or pipeline_vs_reference or probe_batching" > $O/pytest_kcat.txt 2>&1
tail -5 $O/pytest_kcat.txt
python3 tests/diag/ab_cfg.py "LOCO_KCAT=0" "LOCO_KCAT=1" > $O/ab_kcat.txt 2>&1
cat $O/ab_kcat.txt
for k in 0 1; do
LOCO_KCAT=$k python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_kcat$k.json 2> $O/bench_kcat$k.err
python3 -c "
import json; d=json.load(open('$O/bench_kcat$k.json')); r=d['roofline']
print('KCAT=$k', d['ms_per_step'], r['kernel'], r['avg_launch_ms'], d['e2e'])
for kk,v in list(r['all_conv_kernels'].items())[:9]: print('  ', kk, v)
"
done
