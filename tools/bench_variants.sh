#!/bin/bash
# usage: tools/bench_variants.sh "<ENV=.. ENV=..>" "<ENV=..>" ...   — one bench.py run per argument (env assignments), one summary line each
for v in "$@"; do
  out=$(env $v python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | grep '^{')
  python - "$v" "$out" <<'P'
import json, sys
v, out = sys.argv[1], sys.argv[2]
try:
    d = json.loads(out)
    print(f"{v:50s} {d['value']:.4e} updates/s  {d['ms_per_step']:.4f} ms/step  kernel {d['roofline']['avg_launch_ms']:.4f} ms")
except Exception as e:
    print(f"{v:50s} FAILED {e} {out[:200]}")
P
done
