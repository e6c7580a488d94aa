R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3x
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_latent.py -x -q -m gpu -k "flash or ldm_unet" > $O/pytest1.txt 2>&1
grep -E "passed|failed|Error" $O/pytest1.txt | tail -3
python tests/diag/ab_step.py tloco_sd15 > $O/ab.txt 2>&1
cat $O/ab.txt
