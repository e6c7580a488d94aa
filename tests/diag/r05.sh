# Round 5: the gpurun command lists behind profiles/r05_experiments.md, one case per experiment.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tests/diag/r05.sh <case> [args]'
# Cases that time single conv shapes need the diagnostics build (make -C loco-edit_amd/csrc diag; the library must not be listed in
# .gpurunignore for the call); the stamp cases need the stamp build:
#   make -C loco-edit_amd/csrc BUILD=build_stamp EXTRA=-DLOCO_DUAL_STAMP DIAGLIB=../../tests/diag/lib_stamp/libloco_hip_stamp.so diag
cd $GRAFT_REPO_ROOT; W=${1:-help}; shift; O=gpurun_out/r05_$W; mkdir -p $O
DIAG=$GRAFT_REPO_ROOT/loco-edit_amd/libloco_hip_diag.so; STAMP=$GRAFT_REPO_ROOT/tests/diag/lib_stamp/libloco_hip_stamp.so
case $W in
  pmc_mem)      # memory-side + SQ counters of the dominant conv shape: [tag] [prec] [mode] [B] [kernel substring]
    timeout 1500 python3 tests/diag/pmc_conv_mem.py r05_pmc_mem/${1:-run} ${2:-bf16x3} ${3:-3} ${4:-5} ${5:-conv_} ;;
  dual_check)   # the dual-probe tile (or any 0 / 1 switch: $2) against the default, bit for bit: [B] [ENV_NAME]
    timeout 600 python3 tests/diag/dual_check.py ${1:-5} CELEBA_DDPM ${2:-LOCO_CONV_DUAL} | tee $O/check.log ;;
  shapes)       # per-shape times of the 3x3 conv with the dual tile off / on
    for D in 0 1 0 1; do LOCO_HIP_LIB=$DIAG LOCO_CONV_DUAL=$D timeout 300 python3 tests/diag/conv_shapes.py bf16x3 0,3 2>&1 | grep "us " | sed "s/^/dual=$D /"; done | tee $O/shapes.log ;;
  stamps)       # phase stamps + what-if switches: dual tile (dual_stamps.py) and lock-step kernel (lowp_stamps.py)
    for X in 0 2 4 6 14; do LOCO_HIP_LIB=$STAMP LOCO_CONV_DUAL=1 LOCO_DUAL_WHATIF=$X timeout 300 python3 tests/diag/dual_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done | tee $O/dual.log
    for X in 0 2 4 6 14; do LOCO_HIP_LIB=$STAMP LOCO_DUAL_WHATIF=$X timeout 300 python3 tests/diag/lowp_stamps.py 3 128 2>&1 | grep -v amdgpu.ids; done | tee $O/lockstep.log ;;
  ab)           # whole-step A/B of environment configurations: "A=1,B=0" "A=0" ... [-- workload ...]
    timeout 1500 python3 tests/diag/ab_cfg.py "$@" 2>&1 | tee $O/ab.log ;;
  ab_lib)       # whole-step A/B of the libraries under tests/diag/lib (make BUILD=... EXTRA=... LIB=../../tests/diag/lib/libloco_x.so)
    timeout 1500 python3 tests/diag/ab_step.py "$@" 2>&1 | tee $O/ab_lib.log ;;
  gemm_check)   # the DMA-fed 1x1 GEMM against the per-pixel kernel per layer shape: [B] [mode 0|2]
    LOCO_HIP_LIB=$DIAG timeout 600 python3 tests/diag/gemm_check.py ${1:-5} ${2:-0} 2>&1 | grep -v amdgpu.ids | tee $O/gemm_check.log ;;
  kernel_stats) # rocprofv3 --kernel-trace --stats of one bench workload under an environment setting: [workload] [VAR=value]
    cd /tmp && export TMPDIR=/tmp; [ -n "$2" ] && export "$2"
    rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${1:-celeba_top5} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/stats.err
    find $GRAFT_REPO_ROOT/$O -name "*kernel_trace.csv" -delete; head -25 $GRAFT_REPO_ROOT/$O/stats/s_kernel_stats.csv | cut -c1-160 ;;
  attn)         # the attention tangent / cotangent kernels standalone, converting vs DMA-fed, bit comparison + times: [binary ...]
                # (build here first: tests/diag/attn_build.sh [AF_WI bits ...] -> tests/diag/bin/attn_bench[_wi<bits>])
    bash tests/diag/attn_wi.sh "$@"; cp gpurun_out/attn_wi.log $O/ ;;
  *) grep "^  [a-z_]*)" $0 ;;
esac
