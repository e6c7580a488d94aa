# round 5, run 15: bisect the adjointness failure of config 4 with the DMA-fed 1x1 GEMM
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run15; mkdir -p $O
for E in "LOCO_CONV_GEMM=0" "LOCO_CONV_GEMM=1 LOCO_GEMM_NOSPLIT=1" "LOCO_CONV_GEMM=1 LOCO_GEMM_TM=2" "LOCO_CONV_GEMM=1 LOCO_GEMM_TM=2 LOCO_GEMM_NOSPLIT=1"; do
  echo "== $E"; env $E timeout 600 python3 -m pytest tests/test_gpu_latent.py -x -q -s -m gpu -k "config4" 2>&1 | grep "timing\|passed\|failed\|AssertionError: assert\|SD15" | head -12
done | tee $O/bisect.log
