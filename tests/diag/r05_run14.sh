# round 5, run 14: the DMA-fed 1x1 GEMM: transformer parity tests, per-shape times (LOCO_CONV_GEMM=0 / 1), where test_config4 spends its time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run14; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_latent.py -x -q -m gpu -k "spatial_transformer or ldm or transformer" > $O/pytest_xfmr.log 2>&1; tail -4 $O/pytest_xfmr.log
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/loco-edit_amd/libloco_hip_diag.so
for G in 0 1; do LOCO_CONV_GEMM=$G timeout 300 python3 tests/diag/conv1x1_sd.py 2>&1 | grep "1x1" | sed "s/^/gemm=$G /"; done | tee $O/shapes.log
unset LOCO_HIP_LIB
timeout 900 python3 -m pytest tests/test_gpu_latent.py -x -q -s -m gpu -k "config4" > $O/pytest_c4.log 2>&1; grep "timing\|passed\|failed\|SD15" $O/pytest_c4.log
