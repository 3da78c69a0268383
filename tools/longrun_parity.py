import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import conftest
from sphexample_amd.engine import make_engine
from oracle.oracle import make_oracle
p, s = conftest.load_dam_break_3d_shipped()
for fb in (8, 4):
    e = make_engine(p, s, device_float_bytes=fb); o = make_oracle(p, s)
    done = 0
    for chunk in (100, 400, 1000, 1500):
        t0 = time.time(); pe = e.advance(1e9, max_steps=chunk); te = time.time() - t0
        t0 = time.time(); po = o.advance(1e9, max_steps=chunk); to = time.time() - t0
        done += chunk
        a, b = e.download(), o.download()
        ia, ib = np.argsort(a["ID"]), np.argsort(b["ID"])
        dr = np.abs(a["Density"][ia] - b["Density"][ib]).max() / np.abs(b["Density"]).max()
        dx = np.abs(a["Position"][ia] - b["Position"][ib]).max() / np.abs(b["Position"]).max()
        print(f"fp{fb*8} steps {done}: rebuilds {pe.n_rebuilds}/{po.n_rebuilds} t {pe.total_time:.6f}/{po.total_time:.6f} rho {dr:.2e} x {dx:.2e}  gpu {te:.2f}s cpu {to:.2f}s", flush=True)
