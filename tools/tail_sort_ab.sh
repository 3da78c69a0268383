# Round 6: the end of every XCD run ordered again by cost ($SPHMI_TAIL_SORT = per mille of the run; 0 = off): bench windows, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2 3 4; do for ts in 0 200 350 100; do
  SPHMI_TAIL_SORT=$ts python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('tail_sort $ts  value %.4e  ms/step %.4f  kernel ms %.4f' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms']))"
done; done > gpurun_out/r06/tail_sort_ab.txt 2>&1
sort gpurun_out/r06/tail_sort_ab.txt
