R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for rep in 1 2; do
for c in "LOCO_CONV_STAGGER=0" "LOCO_CONV_STAGGER=2" "LOCO_CONV_STAGGER=4" "LOCO_CONV_STAGGER=8" "LOCO_CONV_STAGGER=16"; do
env $c python3 tests/diag/conv_shapes.py bf16x3 0,3 2>&1 | grep "us " >> $O/shapes.txt
done; done
sort $O/shapes.txt > $O/shapes_sorted.txt
cat $O/shapes_sorted.txt
unset LOCO_HIP_LIB
python3 tests/diag/ab_cfg.py "LOCO_CONV_STAGGER=0" "LOCO_CONV_STAGGER=4" "LOCO_CONV_STAGGER=8" > $O/ab.txt 2>&1
cat $O/ab.txt
