# Round 6: the XCD-share feedback on large launches — $SPHMI_XCD_FEEDBACK=2 (every size: rounds 2-5) against the default (below 5 000 tiles only), interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2 3 4; do for fb in 2 1; do for win in "20 5" "100 10"; do set -- $win
  SPHMI_XCD_FEEDBACK=$fb python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('xcd_feedback $fb  steps %3d  value %.4e  ms/step %.4f  kernel ms %.4f' % (j['steps'], j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms']))"
done; done; done > gpurun_out/r06/xcd_feedback_default_ab.txt 2>&1
sort gpurun_out/r06/xcd_feedback_default_ab.txt
