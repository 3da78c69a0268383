#!/bin/bash
# Round profile of bench.py on the GPU box: rocprofv3 --kernel-trace --stats of the bench command (the PMC counter passes are
# tools/pmc_passes.sh — separate runs: gpurun refuses --pmc together with trace domains).
# usage: tools/profile_round.sh <tag> [bench args]   → gpurun_out/prof_<tag>/{kernel_stats.md,bench_line.json}
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
args=${@:---steps 100 --warmup 10 --no-cpu-baseline --no-extras --precondition-ms 0}
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $R/bench.py $args > $out/bench_trace.log 2>&1
grep '^{' $out/bench_trace.log > $out/bench_line.json
db=$(find $out/trace -name '*.db' | head -1)
echo "command: rocprofv3 --kernel-trace --stats -- python bench.py $args" > $out/kernel_stats.md
echo >> $out/kernel_stats.md
python $R/tools/prof_summary.py "$db" >> $out/kernel_stats.md
rm -rf $out/trace
head -12 $out/kernel_stats.md; cut -c1-300 $out/bench_line.json
