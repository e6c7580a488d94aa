# round 5, run 8: probe-batched tangent / cotangent statistics + norm-cotangent apply: bits, whole-step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run8; mkdir -p $O
LOCO_CONV_DUAL=0 timeout 600 python3 tests/diag/dual_check.py 5 CELEBA_DDPM LOCO_TSTATS_PB > $O/pb_check.log 2>&1; tail -4 $O/pb_check.log
LOCO_CONV_DUAL=0 timeout 600 python3 tests/diag/dual_check.py 3 CELEBA_DDPM LOCO_TSTATS_PB > $O/pb_check3.log 2>&1; tail -4 $O/pb_check3.log
timeout 1200 python3 tests/diag/ab_cfg.py "LOCO_CONV_DUAL=0,LOCO_TSTATS_PB=0" "LOCO_CONV_DUAL=0,LOCO_TSTATS_PB=1" "LOCO_CONV_DUAL=1,LOCO_TSTATS_PB=1" -- celeba_top5 2>&1 | tee $O/ab.log
