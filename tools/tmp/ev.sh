cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_examples.py 2000 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launch_time_samples'])"
python bench.py --dp 0.0085 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launch_time_samples'])"
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3
