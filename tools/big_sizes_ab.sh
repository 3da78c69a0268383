cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for dp in 0.003 0.002125; do
  for tree in . build/r3tree; do
    ( cd $tree; echo "### rep $rep dp $dp tree $tree"; timeout 300 python bench.py --dp $dp --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline'].get('kernel_avg_launch_ms', j['roofline'].get('avg_launch_ms')))
" )
  done
done
done
