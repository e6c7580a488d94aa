mkdir -p gpurun_out/r3d
python -m pytest tests -x -q -m gpu > gpurun_out/r3d/pytest.txt 2>&1
tail -5 gpurun_out/r3d/pytest.txt
python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
tail -c 3000 gpurun_out/r3d/bench.json
