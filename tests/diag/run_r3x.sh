R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3x
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_latent.py -x -q -m gpu -k "forward_jvp or ldm_unet or headline or flash or cross_attention or decoder_forward or encoder_engine" > $O/pytest1.txt 2>&1
tail -3 $O/pytest1.txt
python tests/diag/ab_env.py LOCO_DEEP1 0,1 tloco_sd15 tloco_if64 celeba_top5 > $O/ab.txt 2>&1
cat $O/ab.txt
