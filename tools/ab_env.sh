# A/B of environment switches on the reference's example cases: bash tools/ab_env.sh "A=1 B=2|C=3|" [pytest -k expression]   ('|' separates settings; empty = default)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "$2" 2>&1 | tail -5; fi
IFS='|' read -ra SETS <<< "$1"
for rep in 1 2; do for s in "${SETS[@]}"; do echo "== [$s]"; env $s timeout 600 python tools/bench_examples.py 1000 2>&1 | grep -v "^\[" ; done; done > gpurun_out/ab_env.txt 2>&1
grep -E "==|fp" gpurun_out/ab_env.txt
