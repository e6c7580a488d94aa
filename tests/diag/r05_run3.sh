# round 5, run 3: dual tile: bit-exactness with explicit fma forms, phase stamps of the dual units
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run3; mkdir -p $O
timeout 600 python3 tests/diag/dual_check.py 5 > $O/dual_check.log 2>&1; tail -7 $O/dual_check.log
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/tests/diag/lib/libloco_hip_stamp.so
for M in 3 0; do timeout 300 python3 tests/diag/dual_stamps.py $M 128 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps.log
timeout 300 python3 tests/diag/dual_stamps.py 3 256 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.log
