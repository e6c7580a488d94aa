R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3n
cd $R
python - > gpurun_out/r3n/mem.txt 2>&1 <<'PY'
import torch, loco_edit_amd
from loco_edit_amd.config import SD15_UNET, synth_params
from loco_edit_amd.hip import LocoEngine
e = LocoEngine(SD15_UNET, max_batch=1, device=torch.device("cuda:0"))
print("SD15_UNET workspace at max_batch=1: %.2f GB" % (e.workspace_bytes() / 1e9), "flops %.1f GFLOP" % (e.unet_flops() / 1e9))
PY
cat gpurun_out/r3n/mem.txt | tail -3
timeout 900 python bench.py --workload tloco_sd15 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-e2e 2> gpurun_out/r3n/err.txt | tail -1 > gpurun_out/r3n/bench_sd15.json
python -c "import json; d=json.load(open('gpurun_out/r3n/bench_sd15.json')); print('tloco_sd15', d['ms_per_step'], d['value'], d['singular_values'][:3], d['roofline']['kernel'], d['roofline']['frac'])" || tail -5 gpurun_out/r3n/err.txt
