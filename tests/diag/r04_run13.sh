R=$GRAFT_REPO_ROOT; cd $R
LOCO_B1_TILE=4096 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_vs_reference or forward_jvp_vjp or full_size_forward or p2_full_size" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1024 4096 16384; do
echo "LOCO_B1_TILE=$v $(LOCO_B1_TILE=$v python3 tests/diag/fwd_b1_time.py 1 100 2>&1 | grep 'B=')"
done; done
