mkdir -p gpurun_out/r3a
python tests/diag/ab_step.py celeba_top5 > gpurun_out/r3a/ab.txt 2>&1
LOCO_HIP_LIB=$PWD/tests/diag/lib/v0_base.so python tests/shape_profile.py > gpurun_out/r3a/shape_v0.txt 2>&1
LOCO_HIP_LIB=$PWD/tests/diag/lib/v1_deep.so python tests/shape_profile.py > gpurun_out/r3a/shape_v1.txt 2>&1
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py::test_config3_p2_rank20_of_64_probes_at_size > gpurun_out/r3a/pytest.txt 2>&1
tail -5 gpurun_out/r3a/pytest.txt; cat gpurun_out/r3a/ab.txt
