cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for P in bf16x3 f16; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcA_$P -o a --output-format csv -- python3 $R/tests/diag/conv_pmc.py 3 $P > $R/gpurun_out/pmcA_$P.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA -d $R/gpurun_out/pmcB_$P -o b --output-format csv -- python3 $R/tests/diag/conv_pmc.py 3 $P > $R/gpurun_out/pmcB_$P.log 2>&1
tail -2 $R/gpurun_out/pmcA_$P.log $R/gpurun_out/pmcB_$P.log
done
