# Round 6: gpurun command lists, one case per experiment of profiles/r06_experiments.md.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tests/diag/r06.sh <case> [args]'
cd $GRAFT_REPO_ROOT; W=${1:-help}; shift; O=gpurun_out/r06_$W; mkdir -p $O
DIAG=$GRAFT_REPO_ROOT/loco-edit_amd/libloco_hip_diag.so
case $W in
  power)        # the dominant conv on random vs all-zero operands (clock management), bf16x3 and f16, tangent and raw modes
    for Z in 0 3 0 3; do for P in bf16x3 f16; do for M in 3 0; do
      LOCO_HIP_LIB=$DIAG LOCO_BENCH_ZERO=$Z timeout 300 python3 tests/diag/power_check.py $P $M ${1:-1500} 2>&1 | grep " us"; done; done; done | tee $O/power.log ;;
  bench)        # the headline line, short form: [extra bench args]
    timeout 900 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extra "$@" 2>$O/bench.err | tee $O/bench.json ;;
  shapes)       # per-shape times of the 3x3 conv under environment settings: "A=1 B=2" "A=0" ...
    for E in "$@"; do for R in 0 1; do env $E LOCO_HIP_LIB=$DIAG timeout 300 python3 tests/diag/conv_shapes.py bf16x3 0,3,4 2>&1 | grep "us " | sed "s/^/{$E} /"; done; done | tee $O/shapes.log ;;
  wi)           # what-if builds of the 3x3 kernel (tests/diag/libwi/*.so: MFMA shape, two MFMAs per product, no lo fragment reads) vs the diag build
    for R in 0 1; do for L in $DIAG tests/diag/libwi/*.so; do for M in 3 0; do
      LOCO_HIP_LIB=$GRAFT_REPO_ROOT/${L#$GRAFT_REPO_ROOT/} timeout 300 python3 tests/diag/power_check.py bf16x3 $M ${1:-1000} 2>&1 | grep " us" | tail -1 | sed "s|^|$(basename $L) |"; done; done; done | tee $O/wi.log ;;
  ab)           # whole-step A/B of environment configurations: "A=1,B=0" "A=0" ... [-- workload ...]
    timeout 1500 python3 tests/diag/ab_cfg.py "$@" 2>&1 | tee $O/ab.log ;;
  lin)          # the tangent / cotangent means in the conv epilogues: parity tests + A/B
    timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -k "fused_into or dual_probe or persistent_conv or bench_two_ranks" 2>&1 | tail -5 | tee $O/pytest.log
    timeout 900 python3 tests/diag/ab_cfg.py "LOCO_FUSE_LIN=1" "LOCO_FUSE_LIN=0" 2>&1 | tee $O/ab.log ;;
  kstats)       # rocprofv3 --kernel-trace --stats of two timed headline steps per environment setting: "A=1" "A=0" ...
    cd /tmp && export TMPDIR=/tmp
    for E in "$@"; do T=$(echo $E | tr -c 'A-Za-z0-9=\n' '_'); ( export $E
      rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/$T -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-extra --no-profile > $GRAFT_REPO_ROOT/$O/$T.json 2> $GRAFT_REPO_ROOT/$O/$T.err )
      find $GRAFT_REPO_ROOT/$O/$T -name "*kernel_trace.csv" -delete; cp $(find $GRAFT_REPO_ROOT/$O/$T -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/$T.csv; rm -rf $GRAFT_REPO_ROOT/$O/$T; done ;;
  *) grep "^  [a-z_]*)" $0 ;;
esac
