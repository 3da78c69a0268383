# Round 6: the scheduling knobs of rounds 1-4 re-measured on the current kernel (C3 bench, 60 steps after 10, interleaved)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
run() { env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('%-34s value %.4e  kernel ms %.4f' % ('$*', j['value'], j['roofline']['avg_launch_ms']))"; }
for rep in 1 2 3; do
  run A=0; run SPHMI_RESCHED=0; run SPHMI_XCD_SEGS=2; run SPHMI_XCD_SEGS=4; run SPHMI_CLASSES_FINE_BELOW=100000; run SPHMI_TAIL_SORT=0; run SPHMI_TAIL_SORT=350; run SPHMI_TPB2=0; run SPHMI_TPB2=4
done > gpurun_out/r06/sched_knobs_ab.txt 2>&1
sort gpurun_out/r06/sched_knobs_ab.txt
