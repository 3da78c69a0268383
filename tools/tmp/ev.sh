cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('new', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launch_time_samples'])"
(cd build/r3tree; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('r3 ', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])")
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('new100', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['launch_time_samples'])"
timeout 300 python tools/bench_examples.py 2000 2>&1 | grep fp32
bash tools/window_curve.sh 20 5 d
