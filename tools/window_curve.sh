#!/bin/bash
# Kernel time launch by launch through a bench window (is the short window slower because of the device's clock state, or the schedule?)
# usage (gpurun): bash tools/window_curve.sh <steps> <warmup> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/curve_$3
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace -d $out/t -o t -- python $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extras > $out/bench.log 2>&1
db=$(find $out/t -name '*.db' | head -1)
python $R/tools/launch_timeline.py $db 300 | grep k_neighbor_force | tail -n $((2 * ($1 + $2) + 8)) > $out/timeline.txt
grep '^{' $out/bench.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
rm -rf $out/t
