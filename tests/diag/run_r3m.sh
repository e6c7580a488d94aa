R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3m
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_latent.py -x -q -m gpu -k "flash_attention or ldm_unet" -s > gpurun_out/r3m/pytest1.txt 2>&1
grep -E "flash J V|LDM|passed|failed|^E |Error" gpurun_out/r3m/pytest1.txt | tail -20
