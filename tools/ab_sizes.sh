# A/B of build variants over sizes: bash tools/ab_sizes.sh "v1 v2" "dp dp ..." [pytest -k expression]
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
if [ -n "$3" ]; then timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_multi_gpu.py -m gpu -x -q -k "$3" 2>&1 | tail -5; fi
for rep in 1 2; do for v in $1; do echo "== $v"; SPHMI_LIB=build/variants/libsphmi_$v.so timeout 900 python tools/time_sizes.py $2 2>&1 | grep -v "^\["; done; done | tee gpurun_out/ab_sizes.txt
