R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3x
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_tloco.py tests/test_gpu_latent.py -x -q -m gpu -k "not at_size and not at_stable and not config4_on and not encoder_at" > $O/pytest1.txt 2>&1
grep -E "passed|failed|Error" $O/pytest1.txt | tail -3
python tests/diag/ab_env.py LOCO_CFG_STREAMS 0,1 tloco_if64 tloco_sd tloco_sd15 > $O/ab.txt 2>&1
cat $O/ab.txt
