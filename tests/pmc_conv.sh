cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcA -o a --output-format csv -- python3 $R/tests/conv_pmc.py 3 > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA -d $R/gpurun_out/pmcB -o b --output-format csv -- python3 $R/tests/conv_pmc.py 3 > $R/gpurun_out/pmcB.log 2>&1
tail -3 $R/gpurun_out/pmcA.log $R/gpurun_out/pmcB.log
