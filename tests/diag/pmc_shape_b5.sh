R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for M in 3 0 4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $O/m${M}_$C -o p --output-format csv -- python3 $R/tests/diag/conv_pmc.py $M bf16x3 > $O/m${M}_$C.log 2>&1
    echo "mode $M: $(python3 $R/tests/diag/pmc_shape.py $O/m${M}_$C $C)  $(tail -1 $O/m${M}_$C.log)"
  done
done
find $O -name "*.csv" -size +1M -delete
