for dp in 0.012 0.0085 0.0065; do for w in 2 4 8; do
  SPHMI_WPT=$w python bench.py --dp $dp --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp $dp WPT $w N', j['config']['particles'], '%.4g upd/s' % j['value'], 'kernel %.4f ms' % j['roofline']['avg_launch_ms'])"
done; done
