R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
LOCO_CONV_2WG=1 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or headline_config or full_size_forward or statistics_fused" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
python3 tests/diag/ab_cfg.py "LOCO_KCAT=0,LOCO_CONV_2WG=0" "LOCO_KCAT=0,LOCO_CONV_2WG=1" "LOCO_KCAT=1,LOCO_CONV_2WG=0" > $O/ab.txt 2>&1
cat $O/ab.txt
export LOCO_HIP_LIB=$R/loco-edit_amd/libloco_hip_diag.so
for c in "LOCO_CONV_2WG=0" "LOCO_CONV_2WG=1"; do env $c python3 tests/diag/conv_shapes.py bf16x3 0,3 2>&1 | grep "us " ; done
