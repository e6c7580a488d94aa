# round 5, run 17: the DMA-fed 1x1 GEMM in the flow: config 4 test (adjointness), whole-solve A/B of config 4, headline + config 5 sanity
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run17; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_latent.py -x -q -s -m gpu -k "config4 or spatial_transformer or 2_1_base" 2>&1 | grep "timing\|passed\|failed\|AssertionError: assert\|SD15" | tee $O/pytest.log
timeout 1500 python3 tests/diag/ab_cfg.py "LOCO_CONV_GEMM=0" "LOCO_CONV_GEMM=1" "LOCO_CONV_GEMM=1,LOCO_GEMM_ALL=1" -- tloco_sd15 2>&1 | tee $O/ab.log
