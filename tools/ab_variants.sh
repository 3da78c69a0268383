# A/B of build variants on the reference's example cases: bash tools/ab_lone_pairs.sh "v1 v2 ..." [pytest -k expression]
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
VARS=${1:-"nosplit split"}
if [ -n "$2" ]; then timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "$2" 2>&1 | tail -5; fi
for v in $VARS $VARS; do echo "== $v"; SPHMI_LIB=build/variants/libsphmi_$v.so timeout 600 python tools/bench_examples.py 1000 2>&1 | grep -v "^\[" ; done > gpurun_out/ab_examples.txt 2>&1
grep -E "==|fp" gpurun_out/ab_examples.txt
