R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for B in 1 2; do
for M in 0 3; do
  C=FETCH_SIZE
  rocprofv3 --kernel-trace --pmc $C -d $O/b${B}m${M}_$C -o p --output-format csv -- python3 $R/tests/diag/conv_pmc.py $M bf16x3 $B > $O/b${B}m${M}_$C.log 2>&1
  echo "B $B mode $M: $(python3 $R/tests/diag/pmc_shape.py $O/b${B}m${M}_$C $C)"
done
done
find $O -name "*.csv" -delete
