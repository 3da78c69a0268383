for dp in ${DPS:-0.0052 0.0048 0.0045 0.00425}; do for w in 1 2; do
  SPHMI_WPT=$w timeout 120 python bench.py --dp $dp --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('WPT $w dp $dp N', j['config']['particles'], 'tiles', j['config']['particles']//64, '%.4g upd/s' % j['value'], 'kernel %.4f ms' % j['roofline']['avg_launch_ms'])"
done; done
