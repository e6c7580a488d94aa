# SQ counters of the dominant conv shape (tangent 128 -> 128 @256^2, 5 probes) in both low-precision arithmetics, two
# PMC passes each (8 SQ slots per pass), + the kernel-trace average of the same launches (program directly after `--`).
# Needs the diagnostics build (make -C loco-edit_amd/csrc diag): loco_bench_conv.   bash tests/diag/pmc_conv_r04.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r04}; O=$R/gpurun_out/$T; mkdir -p $O
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for P in bf16x3 f16; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $O/pmcA_$P -o a --output-format csv -- python3 $R/tests/diag/conv_pmc.py 3 $P > $O/pmcA_$P.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM -d $O/pmcB_$P -o b --output-format csv -- python3 $R/tests/diag/conv_pmc.py 3 $P > $O/pmcB_$P.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_$P -o t --output-format csv -- python3 $R/tests/diag/conv_pmc.py 3 $P > $O/trace_$P.log 2>&1
python3 $R/tests/diag/pmc_summarise.py $O/pmcA_$P "conv_mfma" > $O/conv3x3_tan_pmc_sq_a_$P.csv
python3 $R/tests/diag/pmc_summarise.py $O/pmcB_$P "conv_mfma" > $O/conv3x3_tan_pmc_sq_b_$P.csv
grep conv_mfma $O/trace_$P/*/*kernel_stats.csv $O/trace_$P/*kernel_stats.csv 2>/dev/null | head -3 > $O/conv3x3_tan_trace_$P.txt
tail -1 $O/pmcA_$P.log $O/trace_$P.log
find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*kernel_trace.csv" -delete
done
cat $O/conv3x3_tan_pmc_sq_a_bf16x3.csv $O/conv3x3_tan_pmc_sq_b_bf16x3.csv $O/conv3x3_tan_trace_*.txt
