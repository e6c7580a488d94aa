# round 5, run 6: what-if timing of the dual tile's loop skeleton (no memory traffic; no conversions; no barrier)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run6; mkdir -p $O
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp_s0.so
for W in 6 14 30 22; do LOCO_DUAL_WHATIF=$W timeout 300 python3 tests/diag/dual_stamps.py 3 128 2>&1 | grep -v amdgpu.ids | grep "whatif\|chunk loop\|unit "; done | tee $O/stamps.log
