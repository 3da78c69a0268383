#!/bin/bash
# Fine (16) or coarse (4) cost classes of the tile order by tile count.  usage: DPS="…" CS="…" tools/classes_threshold.sh
for dp in ${DPS:-0.0048 0.0045 0.00425}; do for c in ${CS:-10000 13000 15000 20000}; do
  SPHMI_CLASSES_FINE_BELOW=$c timeout 120 python bench.py --dp $dp --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fine<$c dp $dp tiles', j['config']['particles']//64, '%.4g upd/s' % j['value'], 'kernel %.4f ms' % j['roofline']['avg_launch_ms'])"
done; done
