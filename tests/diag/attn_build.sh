#!/bin/bash
# Builds the standalone attention harness (tests/diag/attn_bench.hip) with the library's compiler flags into tests/diag/bin/
# (git-ignored, travels with gpurun): no argument -> attn_bench; arguments -> one binary per what-if bit set, attn_bench_wi<bits>
# (AF_WI bits, csrc/attn_flash.hip: 1 no loads of P, 2 no operand transfers after the first block, 4 no fragment conversions,
#  8 converting kernels forced to two workgroups per CU / DMA-fed launch skipped = the split pass alone, 16 one MFMA per product).
cd "$(dirname "$0")"; mkdir -p bin
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp -I../../loco-edit_amd/csrc"
if [ $# -eq 0 ]; then hipcc $FLAGS attn_bench.hip -o bin/attn_bench; fi
for wi in "$@"; do hipcc $FLAGS -DAF_WI=$wi attn_bench.hip -o bin/attn_bench_wi$wi & done; wait
ls -la bin
