# round 5, run 2: dual tile: bit-exactness, per-shape times, SQ counters of the dual kernel (4 probes = dual units only)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run2; mkdir -p $O
timeout 600 python3 tests/diag/dual_check.py 5 > $O/dual_check.log 2>&1; tail -4 $O/dual_check.log
export LOCO_HIP_LIB=$GRAFT_REPO_ROOT/loco-edit_amd/libloco_hip_diag.so
for D in 0 1; do LOCO_CONV_DUAL=$D timeout 300 python3 tests/diag/conv_shapes.py bf16x3 0,3 2>&1 | grep "us " | sed "s/^/dual=$D /" ; done > $O/shapes.log 2>&1
cat $O/shapes.log
unset LOCO_HIP_LIB
PMC_BLOCKS=SQ,GRBM timeout 900 python3 tests/diag/pmc_conv_mem.py r05_run2/pmc_dual bf16x3 3 4 conv_dual > $O/pmc_dual.log 2>&1
LOCO_CONV_DUAL=0 PMC_BLOCKS=SQ,GRBM timeout 900 python3 tests/diag/pmc_conv_mem.py r05_run2/pmc_single bf16x3 3 4 conv_mfma > $O/pmc_single.log 2>&1
paste -d, $O/pmc_dual/conv3x3_tan_pmc_mem_bf16x3.csv $O/pmc_single/conv3x3_tan_pmc_mem_bf16x3.csv
