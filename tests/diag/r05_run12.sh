# round 5, run 12: phase stamps + what-ifs of the lock-step 128 x 256 kernel (the kernel every level runs on)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run12; mkdir -p $O
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp.so
for W in 0 2 4 6 14; do LOCO_DUAL_WHATIF=$W timeout 300 python3 tests/diag/lowp_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.log
