#!/bin/bash
# Mid-size launches (two waves per tile) with and without pairing the tiles into workgroups of four waves.  usage: tools/mid_sweep.sh
for t in 0 1; do for dp in 0.0085 0.0065 0.0055; do
  SPHMI_TPB2=$t timeout 120 python bench.py --dp $dp --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('TPB2=$t dp $dp N', j['config']['particles'], '%.4g upd/s' % j['value'], 'kernel %.4f ms' % j['roofline']['avg_launch_ms'])"
done; done
