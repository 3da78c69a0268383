import os, sys, subprocess
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    from conftest import load_dam_break_2d
    from sphexample_amd.engine import make_engine
    p, s = load_dam_break_2d()
    e = make_engine(p, s, device_float_bytes=4)
    e.advance(1e9, max_steps=50)
    del e
    sys.exit(0)
from sphexample_amd import build
lib='/tmp/libsphmi_stats.so'
build.build(force=True, extra_flags=['-DSPHMI_STATS'], out=lib)
for w in ('4','8'):
    r = subprocess.run([sys.executable, __file__, '--child'], env=dict(os.environ, SPHMI_LIB=lib, SPHMI_WPT=w), capture_output=True, text=True)
    print('WPT', w, [l for l in r.stderr.splitlines() if 'stats' in l])
