#!/bin/bash
# Re-takes the measurements behind profiles/HISTORY.md §4.5 / §4.6 and profiles/r03_* in two stages.
#
#   stage 1 (HERE, no GPU: hipcc cross-compiles):   tools/reproduce_round3.sh build
#   stage 2 (on an MI355X box, e.g. through gpurun): tools/reproduce_round3.sh run [what ...]
#       what = pipe | deep | scanpf | queue | mask | diag | f16 | small | sizes | vsr2 | final     (default: all, ≈30 GPU-minutes)
#
# Every table of profiles/r03_pair_loop_experiments.md names the variant builds it compares; the builds are interleaved by
# tools/bench_libs.py so that box-to-box and clock drift hit all of them alike.  Results land under gpurun_out/repro/.
#
# ROUND 4: most of the switches this script builds were retired from sphmi_kernels.h (profiles/r04_retired_switches.patch).
# Run it on the round-3 tree (`git checkout c32c1f7`), or apply that patch first.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ "${1:-}" = "build" ]; then
  python tools/prebuild_variants.py \
    "pipe0:-DSPHMI_PIPE=0" "pipe1:-DSPHMI_PIPE=1" \
    "deep1:-DSPHMI_DEEP=1" "deep1w6:-DSPHMI_DEEP=1 -DSPHMI_MIN_WAVES=6" "deep3:-DSPHMI_DEEP=3" "deep3w6:-DSPHMI_DEEP=3 -DSPHMI_MIN_WAVES=6" \
    "pf1:-DSPHMI_SCAN_PF=1" "pf2:-DSPHMI_SCAN_PF=2" "pf3:-DSPHMI_SCAN_PF=3" \
    "q11:-DSPHMI_QUEUE=11" "q10p:-DSPHMI_QUEUE=10 -DSPHMI_QUEUE_C=12" "q11p:-DSPHMI_QUEUE=11 -DSPHMI_QUEUE_C=12" \
    "maskstore:-DSPHMI_MASK_STORE=1" "f16scan:-DSPHMI_F16_SCAN=1" \
    "diag1:-DSPHMI_DIAG=1" "diag2:-DSPHMI_DIAG=2" "diag3:-DSPHMI_DIAG=3" "diag4:-DSPHMI_DIAG=4" "diag5:-DSPHMI_DIAG=5" \
    "p2off:-DSPHMI_PIPE2=0" "lds:-DSPHMI_LDS_STAGE=1"
  python tools/prebuild_variants.py "f64h:-DSPHMI_PIPE_F64=1 -DSPHMI_TPB_F64=1" "f64a:-DSPHMI_PIPE_F64=0 -DSPHMI_TPB_F64=1" "f64b:-DSPHMI_PIPE_F64=1 -DSPHMI_TPB_F64=0" "f64c:-DSPHMI_PIPE_F64=0 -DSPHMI_TPB_F64=0"
  # round 2's tree next to this one (its own Python package and library): build/r2tree
  rm -rf build/r2tree && mkdir -p build/r2tree && git archive 015ef3e~1 | tar -x -C build/r2tree && cp tools/bench_variants.py build/r2tree/tools/ \
    && (cd build/r2tree && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wno-unused-function -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form sphexample_amd/csrc/sphmi_engine.hip -o sphexample_amd/libsphmi.so)
  gcc -shared -fPIC -O1 -o build/libaborttrace.so tools/abort_trace.c
  mkdir -p tools/ubench/bin && hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/dep_chain tools/ubench/dep_chain.hip
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/ubench/bin/mask_pack tools/ubench/mask_pack.hip
  exit 0
fi
[ "${1:-}" = "run" ] || { sed -n 2,11p "$0"; exit 1; }
shift
what=${*:-pipe deep scanpf queue mask diag f16 small sizes vsr2 final}
out=gpurun_out/repro; mkdir -p $out
B="python tools/bench_libs.py"
for w in $what; do
  case $w in
    pipe)   $B 3 pipe0 pipe1 > $out/pipe.txt; $B 2 pipe0 pipe1 -- --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/pipe_w20.txt ;;
    deep)   $B 3 pipe1 deep1 deep1w6 deep3 deep3w6 > $out/deep.txt ;;
    scanpf) $B 3 pipe1 pf1 pf2 pf3 > $out/scan_prefetch.txt ;;
    queue)  $B 3 pipe1 q11 q10p q11p > $out/queue_depth.txt ;;
    mask)   python -m pytest tests/test_mask_handover_gpu.py -q > $out/mask_tests.txt 2>&1
            for r in 1 2 3; do for ms in 0 1; do SPHMI_LIB=$ROOT/build/variants/libsphmi_maskstore.so SPHMI_MASK_STORE=$ms python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
              | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask_store=$ms', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"; done; done > $out/mask_handover_bench.txt ;;
    diag)   $B 2 pipe1 diag1 diag2 diag3 diag4 diag5 > $out/diag.txt ;;
    f16)    $B 3 pipe1 f16scan > $out/f16scan.txt
            SPHMI_LIB=$ROOT/build/variants/libsphmi_f16scan.so python -m pytest tests/test_engine_gpu.py -q -k "single_force or k_step or waves_per_tile or cutoff or dam_break" > $out/f16scan_parity.txt 2>&1 ;;
    small)  for v in p2off pipe1; do echo "== $v"; SPHMI_LIB=$ROOT/build/variants/libsphmi_$v.so python tools/bench_examples.py 2000 2>&1 | grep fp32; done > $out/examples_pipe2.txt
            ./tools/ubench/bin/dep_chain > $out/dep_chain.txt
            ./tools/ubench/bin/mask_pack > $out/mask_pack.txt ;;
    sizes)  for dp in 0.0115 0.0085 0.0067 0.0057 0.005 0.003 0.002125; do python bench.py --dp $dp --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null \
              | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp $dp', j['config']['particles'], j['value'], j['ms_per_step'])"; done > $out/sizes.txt ;;
    vsr2)   python tools/bench_variants.py 200 > $out/variants_head.txt 2>/dev/null
            (cd build/r2tree && python tools/bench_variants.py 200 > $ROOT/$out/variants_round2.txt 2>/dev/null)
            for v in f64h f64a f64b f64c; do SPHMI_LIB=$ROOT/build/variants/libsphmi_$v.so python tools/bench_f64.py 100 2>/dev/null; done > $out/f64_switches.txt ;;
    final)  python bench.py > $out/bench.json 2> $out/bench.err
            python bench.py --steps 20 --warmup 5 > $out/bench_driver_window.json 2>/dev/null
            bash tools/profile_round.sh r03 > /dev/null 2>&1
            bash tools/profile_round.sh r03_driver_window --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
            bash tools/pmc_passes.sh r3_pmc_final > $out/pmc_passes_final.txt 2>&1
            python tools/pmc_derive.py $out/pmc_passes_final.txt gpurun_out/r3_pmc_final/kernel_identity.json $out/counters.json "re-taken by tools/reproduce_round3.sh"
            python tools/bench_examples.py 2000 > $out/examples.txt 2>&1 ;;
    *) echo "unknown: $w" ;;
  esac
  echo "done: $w"
done
