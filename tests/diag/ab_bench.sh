#!/bin/bash
# Same-call A/B of library builds (by hand, through gpurun): bash tests/diag/ab_bench.sh libA.so libB.so ...
# Every library in loco-edit_amd/ named on the command line runs the default bench twice, interleaved.
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do
  for l in "$@"; do
    printf "%-28s " $l
    LOCO_HIP_LIB=$PWD/loco-edit_amd/$l timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
