R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or full_size_forward_vs_golden or headline_config or statistics_fused" > $O/pytest_spec.txt 2>&1
tail -5 $O/pytest_spec.txt
LOCO_SPEC_DMA=0 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or headline_config" > $O/pytest_spec_dma0.txt 2>&1
tail -3 $O/pytest_spec_dma0.txt
python3 tests/diag/ab_cfg.py "LOCO_CONV_SPEC=0" "LOCO_CONV_SPEC=1,LOCO_SPEC_DMA=0" "LOCO_CONV_SPEC=1,LOCO_SPEC_DMA=1" > $O/ab_spec.txt 2>&1
cat $O/ab_spec.txt
for c in "LOCO_CONV_SPEC=0" "LOCO_CONV_SPEC=1"; do
env $c python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $O/bench_$c.json 2> $O/bench_$c.err
python3 -c "
import json; d=json.load(open('$O/bench_$c.json')); r=d['roofline']
print('$c', d['ms_per_step'], d['clock']['sclk_mhz_avg_over_timed_region'], r['kernel'], r['avg_launch_ms'], r['frac'])
for k,v in list(r['all_conv_kernels'].items())[:8]: print('  ', k, v)
"
done
