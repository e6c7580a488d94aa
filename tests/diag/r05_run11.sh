# round 5, run 11: dual tile with explicit counted vmcnt / lgkmcnt waits (asm loads): bits, phase stamps, per-shape times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run11; mkdir -p $O
timeout 600 python3 tests/diag/dual_check.py 5 > $O/dual_check.log 2>&1; tail -4 $O/dual_check.log
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp.so
for W in 0 6 14; do LOCO_DUAL_WHATIF=$W timeout 300 python3 tests/diag/dual_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.log
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/loco-edit_amd/libloco_hip_diag.so
for D in 0 1 0 1; do LOCO_CONV_DUAL=$D timeout 300 python3 tests/diag/conv_shapes.py bf16x3 0,3 2>&1 | grep "us " | sed "s/^/dual=$D /" ; done > $O/shapes.log 2>&1
cat $O/shapes.log
