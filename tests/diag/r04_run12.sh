R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "adjointness_and_linearity_full_size or headline_config or forward_jvp_vjp or two_stream_probe_groups_match or p2_full_size or adm_solver or probe_batching" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python3 tests/diag/ab_cfg.py "LOCO_FUSE_COT=0" "LOCO_FUSE_COT=1" > $O/ab.txt 2>&1
cat $O/ab.txt
