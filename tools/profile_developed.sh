#!/bin/bash
# rocprofv3 kernel-trace of the developed-flow run (tools/bench_developed.py: 0 → 0.4 s, then a 200-step window).
# usage (GPU box): tools/profile_developed.sh <tag>   → gpurun_out/prof_<tag>_developed/
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_${tag}_developed
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats -d $out/trace -o dev -- python $R/tools/bench_developed.py 0.4 200 > $out/bench_developed.log 2>&1
db=$(find $out/trace -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" > $out/kernel_stats.md
grep -v amdgpu.ids $out/bench_developed.log | tail -12 > $out/bench_developed.txt
find $out -name '*.db' -size +20M -delete
find $out -name '*.csv' -size +20M -delete
head -12 $out/kernel_stats.md; cat $out/bench_developed.txt
