# one-slab overhead of the slab driver against the plain engine, same box, interleaved (round 5)
for i in 1 2 3; do
for mode in "" "--force-distributed"; do
python bench.py --steps 64 --warmup 10 --no-extras --no-cpu-baseline --precondition-ms 0 $mode 2>/dev/null | grep '^{' | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('${mode:-plain}', '%.4e upd/s  %.4f ms/step  kernel %.4f ms  rebuilds %d (%.3f ms)' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['rebuilds_in_window'], j['rebuild_ms_in_window']))"
done; done
