# round 5, run 5: dual tile with the two wave groups' halo schedules one tap apart (DU_STAGGER) vs in lock step: phase stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run5; mkdir -p $O
timeout 600 python3 tests/diag/dual_check.py 5 > $O/dual_check.log 2>&1; tail -4 $O/dual_check.log
for S in 0 1; do
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp_s$S.so
for W in 0 6; do echo "== stagger $S"; LOCO_DUAL_WHATIF=$W timeout 300 python3 tests/diag/dual_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done
done | tee $O/stamps.log
