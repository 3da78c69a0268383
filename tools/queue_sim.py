#!/usr/bin/env python3
"""Queue simulation behind the depth of the per-lane mask queues of k_neighbor_force (profiles/HISTORY.md §4.4).  CPU only.

For a sample of tiles of the generated 3-D dam break (dp = 0.0085) it rebuilds what phase 1 pushes — per lane, the bit
counts of the non-empty 32-candidate accept masks, in scan order — and replays the kernel's policy (scan until a queue
holds QCAP - 1 entries, run the pair loop until no lane holds more than QCAP - 1 - SLACK, drain at the end) for several
(QCAP, SLACK).  Output: pair-loop iterations per tile and the fraction of lane slots that do a pair.  The loop counters of
a -DSPHMI_STATS build (71.7 % at QCAP 8 / SLACK 4) match the simulated 70.1 %.
usage: python tools/queue_sim.py [tiles]"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
dp=0.0085
p,s=dam_break_3d(dp),setup_dam_break_3d(dp)
H=s.SimKernel.H
x=p.Position
c=(np.sign(x)*np.trunc(np.abs(x)/H+0.5)).astype(np.int64)
gmin=c.min(0); c=c-gmin+1; npd=c.max(0)+2
key=c[:,0]+npd[0]*(c[:,1]+npd[1]*c[:,2])
o=np.argsort(key,kind='stable'); x=x[o]; key=key[o]
ncell=int(np.prod(npd))
cstart=np.zeros(ncell+2,dtype=np.int64); np.add.at(cstart,key+1,1); cstart=np.cumsum(cstart)
N=len(x); nt=(N+63)//64
rng=np.random.default_rng(0)
tiles=rng.choice(nt-1,size=int(sys.argv[1]) if len(sys.argv)>1 else 200,replace=False)
nxp=npd[0]; nxyp=npd[0]*npd[1]
def tile_masks(t):
    """list of (per-lane 32-bit popcounts) per half-chunk in scan order"""
    a=np.arange(t*64,min(t*64+64,N)); L=len(a)
    out=[]
    for seg in range(9):
        off=((seg%3)-1)*nxp+((seg//3)-1)*nxyp
        lo_l=cstart[key[a]+off-1]; hi_l=cstart[key[a]+off+2]
        LO=lo_l[0]; HI=hi_l[-1]
        for cb in range(LO,HI,64):
            if not ((lo_l<cb+64)&(hi_l>cb)).any(): continue
            cand=np.arange(cb,min(cb+64,HI))
            d=x[a][:,None,:]-x[cand][None,:,:]
            acc=((d*d).sum(2)<=H*H)&(cand[None,:]>=lo_l[:,None])&(cand[None,:]<hi_l[:,None])
            for h in range(2):
                seg_=acc[:,32*h:32*h+32]
                cnt=seg_.sum(1)
                full=np.zeros(64,dtype=np.int64); full[:L]=cnt
                out.append(full)
            out.append(None)   # chunk boundary marker (fullness test happens per chunk)
    return out
def simulate(entries,QCAP=8,SLACK=4):
    q=[[] for _ in range(64)]       # per lane list of remaining bit counts
    cur=np.zeros(64,dtype=np.int64)
    iters=0; lane_it=0
    def burst(keep,drain):
        nonlocal iters,lane_it,cur
        while True:
            ql=np.array([len(v) for v in q])
            owes=(ql>keep) | ((cur>0)|(ql>0) if drain else False)
            if not owes.any(): break
            # one iteration: lanes with cur==0 and queue non-empty refill; lanes with cur>0 consume one
            for l in range(64):
                if cur[l]==0 and q[l]: cur[l]=q[l].pop(0)
            act=cur>0
            cur[act]-=1
            iters+=1; lane_it+=int(act.sum())
    pending=[]
    for e in entries:
        if e is None:
            continue
        pending.append(e)
        if len(pending)==2:
            ql=np.array([len(v) for v in q])
            if (ql>QCAP-2).any(): burst(QCAP-1-SLACK,False)
            for h in pending:
                for l in np.nonzero(h)[0]: q[l].append(int(h[l]))
            pending=[]
    burst(0,True)
    return iters,lane_it
tot={}
per=[]
for t in tiles:
    en=tile_masks(t)
    work=sum(e for e in en if e is not None)
    per.append((work.max(),work.mean()))
    for name,kw in (("Q8S4",dict(QCAP=8,SLACK=4)),("Q8S1",dict(QCAP=8,SLACK=1)),("Q10S1",dict(QCAP=10,SLACK=1)),("Q12S1",dict(QCAP=12,SLACK=1)),("Q12S3",dict(QCAP=12,SLACK=3)),("Q16S1",dict(QCAP=16,SLACK=1)),("Q32S1",dict(QCAP=32,SLACK=1))):
        it,li=simulate(en,**kw)
        a=tot.setdefault(name,[0,0]); a[0]+=it; a[1]+=li
per=np.array(per)
print("tiles",len(tiles),"mean lane work",per[:,1].mean(),"mean of max",per[:,0].mean(),"bound util",per[:,1].sum()/per[:,0].sum())
for k,(it,li) in tot.items(): print(k,"iters/tile",it/len(tiles),"util",li/(64*it))
