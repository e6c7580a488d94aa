#!/bin/bash
# Diagnostic (by hand, CPU only): registers, scratch (spills), LDS and the occupancy the register allocation leaves for every kernel
# of the library, from the compiler's own remarks (-Rpass-analysis=kernel-resource-usage), one line per kernel.
#   bash tests/diag/kernel_resources.sh [out.txt]        (same flags as csrc/Makefile; ~10 min on 8 cores)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-$ROOT/gpurun_out/kernel_resources.txt}
TMP=$(mktemp -d)
cd "$ROOT/loco-edit_amd/csrc"
one() {
  f=$1; sched=max-ilp; [ "$f" = conv_bf16_inst_k.hip ] && sched=max-memory-clause
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -mllvm -amdgpu-sched-strategy=$sched --cuda-device-only \
        -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - - - |
    sed "s/^/$f\t/" > "$2/$f.txt"
}
export -f one
ls *.hip | xargs -P 8 -I{} bash -c "one {} $TMP"
FILT=$(command -v c++filt || command -v llvm-cxxfilt || echo cat)
cat "$TMP"/*.txt | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/' |
  awk -F'\t' '{printf "%s\t%s\t%s\t%s\t%s\t%s\t%s\n", $1, $3, $4, $5, $6, $7, $2}' | $FILT | cut -c1-260 > "$OUT"
rm -rf "$TMP"
echo "wrote $OUT: $(wc -l < "$OUT") kernels, $(grep -vc 'scratch: 0' "$OUT") with scratch"
