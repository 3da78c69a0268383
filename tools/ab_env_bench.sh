# A/B of environment switches on the bench (C3, 60 steps): bash tools/ab_env_bench.sh "A=1|B=2|" <out file under gpurun_out/r05/>   ('|' separates settings; empty = default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
IFS='|' read -ra SETS <<< "$1"
for rep in 1 2 3; do for s in "${SETS[@]}" ""; do
  echo -n "[$s] "; env $s python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.4e upd/s  kernel %.4f ms' % (j['value'], j['roofline']['avg_launch_ms']))"
done; done > gpurun_out/r05/$2 2>&1
sort gpurun_out/r05/$2
