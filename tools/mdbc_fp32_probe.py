import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import conftest, numpy as np
from conftest import perturbed
from sphexample_amd.engine import make_engine
from oracle.oracle import make_oracle
for name in ("dam_break_2d_mdbc","still_wedge_middle_square","duckling","still_wedge"):
    p,s=getattr(conftest,"load_"+name)()
    q=perturbed(p,seed=8,vel_scale=0.05)
    e=make_engine(q,s,device_float_bytes=4); o=make_oracle(q,s)
    e.forces_once(apply_mdbc=True); o.forces_once(apply_mdbc=True)
    a,b=e.download(),o.download()
    ia,ib=np.argsort(a["ID"]),np.argsort(b["ID"])
    err=np.abs(a["Density"][ia]-b["Density"][ib])/1000
    bnd=b["Type"][ib]!=1
    print(name, "max", err.max(), "n>1e-4", (err>1e-4).sum(), "n>1e-5", (err>1e-5).sum(), "of bnd", bnd.sum(), "p99", np.percentile(err[bnd],99))
    k=np.argsort(-err)[:5]
    print("   worst rho gpu/orc:", a["Density"][ia][k], b["Density"][ib][k])
    if name == "dam_break_2d_mdbc":
        ids = b["ID"][ib][k]
        print("   ids", ids, "ghost", b["GhostPoints"][ib][k] if "GhostPoints" in b else None, "pos", b["Position"][ib][k])
