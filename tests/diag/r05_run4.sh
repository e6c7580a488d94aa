# round 5, run 4: what-if timing of the dual tile: halo loads / weight DMAs collapsed onto one address (stamp build)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run4; mkdir -p $O
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp.so
for W in 0 2 4 6; do LOCO_DUAL_WHATIF=$W timeout 300 python3 tests/diag/dual_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.log
