R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3h
cd $R
for m in 1 2 1 2; do LOCO_STREAMS=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-e2e --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$m', d['ms_per_step'], d['parity']['cos_min'])"; done | tee gpurun_out/r3h/streams.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r3h/pytest.txt 2>&1
tail -4 gpurun_out/r3h/pytest.txt
