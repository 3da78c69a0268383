#!/bin/bash
# Small and mid-size cases on the GPU box: the example layouts (µs per step) and the generated dam break at four sizes.
# usage: tools/size_sweep.sh [tag]    (run from the repository root)
python tools/bench_examples.py 1000 2>&1 | grep -E "fp32|fp64" | cut -c1-150
for dp in 0.0085 0.0065 0.0055; do
  python bench.py --dp $dp --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp', '$dp', 'N', j['config']['particles'], '%.4g upd/s' % j['value'], '%.3f ms/step' % j['ms_per_step'], 'kernel %.4f ms' % j['roofline']['avg_launch_ms'])"
done
