R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3l
cd $R
python -m pytest tests/test_gpu_latent.py -x -q -m gpu -k "ldm_unet" -s > gpurun_out/r3l/pytest1.txt 2>&1
grep -E "LDM|passed|failed|^E |Error" gpurun_out/r3l/pytest1.txt | tail -20
