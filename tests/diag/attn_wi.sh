#!/bin/bash
# the attention tangent / cotangent kernels standalone (tests/diag/attn_bench.hip): converting kernels vs the DMA-fed ones, per shape;
# arguments: binaries under tests/diag/bin (default attn_bench); ATTN_SHAPES overrides the shape list (';'-separated)
mkdir -p gpurun_out
IFS=';' read -ra SHAPES <<< "${ATTN_SHAPES:-4096 8 40 5 0;1024 8 80 5 0;1024 6 64 5 128;256 9 64 5 128;1024 4 64 3 0;4096 8 40 3 0}"
for bin in "${@:-attn_bench}"; do
  for shape in "${SHAPES[@]}"; do
    echo "== $bin shape $shape"
    timeout 120 tests/diag/bin/$bin $shape 10
  done
done 2>&1 | tee gpurun_out/attn_wi.log
