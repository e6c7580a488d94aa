R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3x
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_latent.py tests/test_gpu_tloco.py -x -q -m gpu -k "ldm_unet or cross_attention or xattn or config4 or latent_solver_at or with_text" > $O/pytest1.txt 2>&1
grep -E "rel err|passed|failed|Error" $O/pytest1.txt | tail -12
python tests/diag/ab_env.py LOCO_FUSE_XATTN 0,1 tloco_sd15 tloco_sd > $O/ab.txt 2>&1
cat $O/ab.txt
