# A/B of $SPHMI_SIDE_PRIORITY (the slab driver's exchange stream at the greatest priority) on the weak-scaling workloads, slabs of one handle on ONE GPU:
# each slab's two passes alone on the chip (tools/slab_pass_time.py child) and all slabs together (ms per step).   bash tools/side_priority_ab.sh > gpurun_out/side_priority_ab.txt
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in 4 8; do for pr in 0 1; do
  echo "== slabs $w priority $pr rep $rep"
  SPHMI_SIDE_PRIORITY=$pr SPHMI_DD_ONE_SLAB_AT_A_TIME=1 python tools/slab_pass_time.py child $w 40 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = [(q['pass1_alone_ns'] + q['pass2_alone_ns']) / 1e3 for q in r['per_slab']]
print('one slab at a time: mean %.1f us, slowest %.1f us per step' % (sum(t) / len(t), max(t)))"
  SPHMI_SIDE_PRIORITY=$pr python tools/slab_pass_time.py child $w 40 | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('all slabs on the one GPU together: %.1f us per step' % (r['ms_per_step'] * 1e3))"
done; done; done
