R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3x
cd $R
for p in none fwd solve32; do python tests/diag/if_steps.py $p 2>/dev/null | tail -1; done
python -m pytest tests/test_gpu_tloco.py -x -q -m gpu -k "side_by_side or pieces or cfg_operator" 2>&1 | tail -1
python bench.py > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err
python -c "import sys,json; d=json.loads(open('gpurun_out/r3x/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], {k:v.get('ms_per_step') for k,v in d['extra_workloads'].items()})"
