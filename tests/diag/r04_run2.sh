# role-split conv kernel: parity at size, then same-box A/B against the lock-step kernel (LOCO_CONV_SPEC=0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or full_size_forward_vs_golden or headline_config or p2_full_size or statistics_fused or probe_batching or forward_jvp_vjp" > $O/pytest_spec.txt 2>&1
tail -15 $O/pytest_spec.txt
python3 tests/diag/ab_env.py LOCO_CONV_SPEC 0,1 celeba_top5 > $O/ab_spec.txt 2>&1
cat $O/ab_spec.txt
LOCO_CONV_SPEC=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_spec.json 2> $O/bench_spec.err
python3 -c "
import json; d=json.load(open('$O/bench_spec.json')); r=d['roofline']
print(d['ms_per_step'], d['clock'], r['kernel'], r['avg_launch_ms'], r['frac'])
for k,v in list(r['all_conv_kernels'].items())[:12]: print(k, v)
"
