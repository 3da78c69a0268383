#!/bin/bash
# Round profile of bench.py on the GPU box: kernel-trace stats + the two HBM-traffic PMC passes (separate runs, no
# tracing options together with --pmc).  usage: tools/profile_round.sh <tag>   → gpurun_out/prof_<tag>/
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --precondition-ms 0 > $out/bench_trace.log 2>&1
grep '^{' $out/bench_trace.log > $out/bench_line.json
db=$(find $out/trace -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" > $out/kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c -d $out/pmc_$n -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --precondition-ms 0 > $out/pmc_$n.log 2>&1
  db=$(find $out/pmc_$n -name '*.db' | head -1)
  python $R/tools/pmc_summary.py "$db" k_neighbor_force > $out/pmc_$n.txt
done
find $out -name '*.db' -size +20M -delete
cat $out/kernel_stats.md | head -8; cat $out/pmc_*.txt; cat $out/bench_line.json | cut -c1-300
