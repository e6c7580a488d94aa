R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forward_jvp_vjp or full_size_forward_vs_golden or statistics_fused or pipeline_vs_reference or eta1 or p2_full_size or graph_replay" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for f in 1 0; do
LOCO_FUSE_STATS=$f python3 tests/diag/fwd_b1_time.py 1 50 2>&1 | grep "B="
LOCO_FUSE_STATS=$f python3 tests/diag/decode_b25.py 2>&1 | grep "B="
done
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['e2e'])"
