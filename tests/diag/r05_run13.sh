# round 5, run 13: s_setprio 1 for the later-dispatched half of the workgroup in the lock-step stage loop (guide: two waves per SIMD, item 4)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run13; mkdir -p $O
LOCO_CONV_DUAL=0 timeout 1200 python3 tests/diag/ab_step.py celeba_top5 2>&1 | tee $O/ab.log
