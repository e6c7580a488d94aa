R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3y
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_latent.py -x -q -m gpu -k "forward_jvp or ldm_unet or headline or flash or cross_attention or decoder_forward or encoder_engine or config4_on_the_stable" > $O/pytest1.txt 2>&1
grep -E "rel err|finite difference|passed|failed" $O/pytest1.txt | tail -8
bash tests/diag/run_r3w.sh
