cd $GRAFT_REPO_ROOT
echo "=== parity subset"; timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "single_force or k_step or long_run" 2>&1 | tail -3
for rep in 1 2; do for pool in 0 0.03 0.06 0.12; do
echo -n "pool $pool: "; SPHMI_POOL=$pool python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done; done
for pool in 0 0.06 0.12; do echo "pool $pool developed:"; SPHMI_POOL=$pool python tools/bench_developed.py 0.4 200 2>&1 | head -2; done
