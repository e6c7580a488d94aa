R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for rep in 1 2; do
for c in "LOCO_CONV_SPEC=0" "LOCO_CONV_SPEC=1 LOCO_SPEC_DMA=0" "LOCO_CONV_SPEC=1 LOCO_SPEC_DMA=1"; do
env $c python3 tests/diag/conv_shapes.py 2>&1 | grep "us " >> $O/shapes.txt
done; done
sort $O/shapes.txt | awk '{print}' > $O/shapes_sorted.txt
cat $O/shapes_sorted.txt
