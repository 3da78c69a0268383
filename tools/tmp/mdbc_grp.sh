cd $GRAFT_REPO_ROOT
echo "=== mdbc tests default"; timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_multi_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "mdbc or fuzz or ghost" 2>&1 | tail -4
echo "=== mdbc tests group forced"; SPHMI_MDBC_GROUP=1 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -k "mdbc or fuzz" 2>&1 | tail -4
timeout 300 python tools/bench_examples.py 2000 2>&1
