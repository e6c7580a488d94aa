# round 5, run 7: whole-step A/B of the dual tile (headline, p2_k64), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run7; mkdir -p $O
timeout 1200 python3 tests/diag/ab_cfg.py "LOCO_CONV_DUAL=0" "LOCO_CONV_DUAL=1" "LOCO_CONV_DUAL=1,LOCO_DUAL_MIN_UNITS=100" -- celeba_top5 p2_k64 2>&1 | tee $O/ab.log
