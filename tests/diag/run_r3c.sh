mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3c/pytest_parity.txt 2>&1
tail -3 gpurun_out/r3c/pytest_parity.txt
python tests/diag/ab_step.py celeba_top5 > gpurun_out/r3c/ab.txt 2>&1
cat gpurun_out/r3c/ab.txt
LOCO_HIP_LIB=$PWD/tests/diag/lib/e1_epi_lds.so python tests/shape_profile.py > gpurun_out/r3c/shape_e1.txt 2>&1
head -14 gpurun_out/r3c/shape_e1.txt
python -m pytest tests/test_gpu_tloco.py tests/test_gpu_latent.py -x -q -m gpu > gpurun_out/r3c/pytest_tloco.txt 2>&1
tail -3 gpurun_out/r3c/pytest_tloco.txt
