R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3v
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_latent.py -x -q -m gpu -k "ldm_unet or cli_shipped" > $O/pytest1.txt 2>&1
tail -3 $O/pytest1.txt
python tests/diag/ab_env.py LOCO_TILE1_MAXBLK 0,400,1024,4096 tloco_sd15 tloco_if64 celeba_top5 > $O/ab.txt 2>&1
cat $O/ab.txt
