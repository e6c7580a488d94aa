R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flash_attention" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_latent.py -m gpu -q -x -k "config4_on_the_stable or ldm_unet" >> $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 tests/diag/ab_cfg.py "LOCO_FLASH_WIDE=0" "LOCO_FLASH_WIDE=1" -- tloco_sd15 > $O/ab.txt 2>&1; cat $O/ab.txt
