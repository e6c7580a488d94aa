mkdir -p gpurun_out/r3b
python tests/diag/ab_step.py celeba_top5 > gpurun_out/r3b/ab.txt 2>&1
LOCO_HIP_LIB=$PWD/tests/diag/lib/w1_epi_x4_fake.so python tests/shape_profile.py > gpurun_out/r3b/shape_w1.txt 2>&1
cat gpurun_out/r3b/ab.txt; head -12 gpurun_out/r3b/shape_w1.txt
