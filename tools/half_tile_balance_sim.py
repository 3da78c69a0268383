#!/usr/bin/env python3
"""How evenly do the two lanes of a target share its pairs?  (round 5, after the launch became gather-bound: every idle lane slot of a pair
iteration is texture-path time.)  CPU only.  For a sample of half tiles (32 targets, one wave) of the generated 3-D dam break it rebuilds the
accept masks chunk by chunk in scan order and counts the pairs each of the 64 lanes gets under
  interleave  the shipped split: the lane of half h takes candidates 8g + 4h + k of every 32-candidate block (what the matrix layout hands it)
  halves      the lower / upper half of the SET bits of every chunk (contiguous runs: lines stay with one lane), odd counts alternating
  ideal       half of the target's total
and replays the queue policy (scan until a queue holds QCAP - 1 entries, pair loop until no lane holds more than QCAP - 2, drain at the end).
Output: pair-loop iterations per half tile and the fraction of lane slots that do a pair.
usage: python tools/half_tile_balance_sim.py [half tiles] [dp] [jitter]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0085
jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
H = s.SimKernel.H
rng = np.random.default_rng(0)
x = p.Position + jit * dp * rng.standard_normal(p.Position.shape)
c = (np.sign(x) * np.trunc(np.abs(x) / H + 0.5)).astype(np.int64)
gmin = c.min(0); c = c - gmin + 1; npd = c.max(0) + 2
key = c[:, 0] + npd[0] * (c[:, 1] + npd[1] * c[:, 2])
o = np.argsort(key, kind='stable'); x = x[o]; key = key[o]
cstart = np.zeros(int(np.prod(npd)) + 2, dtype=np.int64); np.add.at(cstart, key + 1, 1); cstart = np.cumsum(cstart)
N = len(x); nh = (N + 31) // 32
halves = rng.choice(nh - 1, size=nsample, replace=False)
nxp = npd[0]; nxyp = npd[0] * npd[1]


def chunk_masks(ht):
    """per chunk in scan order: bool [32 targets, 64 candidates]"""
    a = np.arange(ht * 32, min(ht * 32 + 32, N)); out = []
    for seg in range(9):
        off = ((seg % 3) - 1) * nxp + ((seg // 3) - 1) * nxyp
        lo_l = cstart[key[a] + off - 1]; hi_l = cstart[key[a] + off + 2]
        LO = lo_l[0]; HI = hi_l[-1]
        for cb in range(LO, HI, 64):
            if not ((lo_l < cb + 64) & (hi_l > cb)).any(): continue
            cand = np.arange(cb, cb + 64)
            ok = cand < HI
            d = x[a][:, None, :] - x[np.minimum(cand, N - 1)][None, :, :]
            acc = ((d * d).sum(2) <= H * H) & ok[None, :] & (cand[None, :] >= lo_l[:, None]) & (cand[None, :] < hi_l[:, None]) & (cand[None, :] != a[:, None])
            m = np.zeros((32, 64), bool); m[:len(a)] = acc
            out.append(m)
    return out


def lane_entries(masks, mode):
    """per chunk: list of per-lane bit counts of the (one or two) entries the lanes push; lanes 0-31 = half 0, 32-63 = half 1"""
    ent = []
    flip = np.zeros(32, dtype=np.int64)
    for m in masks:
        grp = (np.arange(64) >> 2) & 1
        if mode == "interleave":
            e = np.concatenate([m[:, grp == 0].sum(1), m[:, grp == 1].sum(1)])
        elif mode == "swap":
            # no splitting: per chunk the lane that is behind takes the LARGER of the two interleaved shares (whole words change hands: one entry per lane and chunk as now)
            c0, c1 = m[:, grp == 0].sum(1), m[:, grp == 1].sum(1)
            big, small = np.maximum(c0, c1), np.minimum(c0, c1)
            lane0_big = flip <= 0                                  # flip = T0 - T1 so far
            e0 = np.where(lane0_big, big, small); e1 = np.where(lane0_big, small, big)
            flip = flip + e0 - e1
            ent.append(np.concatenate([e0, e1]))
            continue
        elif mode == "adaptive":
            # as greedy, the cut chosen per chunk among 8 / 16 / ... / 56 so that the totals end up closest
            a0, a1 = m[:, grp == 0], m[:, grp == 1]
            c0, c1 = a0.sum(1), a1.sum(1)
            D = flip + c0 - c1
            best = np.abs(D); give = np.zeros(32, dtype=np.int64)        # signed: > 0 lane 0 hands over, < 0 lane 1
            for cut in (8, 16, 24, 32, 40, 48, 56):
                g0 = m[:, (grp == 0) & (np.arange(64) >= cut)].sum(1); g1 = m[:, (grp == 1) & (np.arange(64) >= cut)].sum(1)
                for sign, g in ((1, g0), (-1, g1)):
                    r = np.abs(D - 2 * sign * g); take = (r < best) & (g > 0) & (np.sign(D) == sign)
                    best = np.where(take, r, best); give = np.where(take, sign * g, give)
            e0 = c0 - np.maximum(give, 0); e1 = c1 - np.maximum(-give, 0)
            flip = D - 2 * give
            ent.append(np.concatenate([e0, e1])); ent.append(np.concatenate([np.maximum(-give, 0), np.maximum(give, 0)]))
            continue
        elif mode.startswith("greedy"):
            # running difference D = T0 - T1 per target; the richer lane hands over the part of ITS mask that lies in candidates >= cut of the chunk
            # (whole groups of four: lines stay with one lane) when that brings the totals closer
            cut = int(mode[6:])
            a0, a1 = m[:, grp == 0], m[:, grp == 1]
            c0, c1 = a0.sum(1), a1.sum(1)
            g0, g1 = m[:, (grp == 0) & (np.arange(64) >= cut)].sum(1), m[:, (grp == 1) & (np.arange(64) >= cut)].sum(1)
            D = flip + c0 - c1
            give0 = (D > 0) & (g0 > 0) & (g0 < D)          # lane 0 richer: hands g0 to lane 1
            give1 = (D < 0) & (g1 > 0) & (g1 < -D)
            e0 = c0 - np.where(give0, g0, 0); e1 = c1 - np.where(give1, g1, 0)
            x0 = np.where(give1, g1, 0); x1 = np.where(give0, g0, 0)       # extra entries received
            flip = D - 2 * np.where(give0, g0, 0) + 2 * np.where(give1, g1, 0)
            ent.append(np.concatenate([e0, e1])); ent.append(np.concatenate([x0, x1]))
            continue
        else:
            n = m.sum(1)
            if mode == "halves":
                k0 = (n + flip) // 2; flip = np.where(n % 2 == 1, 1 - flip, flip)
                e = np.concatenate([k0, n - k0])
            else:
                e = np.concatenate([n / 2.0, n / 2.0])
        ent.append(e)
    return ent


def simulate(ent, QCAP=10):
    q = [[] for _ in range(64)]; cur = np.zeros(64); iters = 0; slots = 0.0

    def burst(keep, drain):
        nonlocal iters, slots, cur
        while True:
            ql = np.array([len(v) for v in q])
            if not ((ql > keep).any() if not drain else ((cur > 0) | (ql > 0)).any()): break
            for l in range(64):
                if cur[l] <= 0 and q[l]: cur[l] = q[l].pop(0)
            act = cur > 0
            slots += np.minimum(cur[act], 1.0).sum(); cur[act] -= 1; cur = np.maximum(cur, 0); iters += 1
    for e in ent:
        if max(len(v) for v in q) > QCAP - 2: burst(QCAP - 2, False)
        for l in range(64):
            if e[l] > 0: q[l].append(float(e[l]))
    burst(0, True)
    return iters, slots


for mode in ("interleave", "halves", "swap", "greedy32", "adaptive"):
    it = 0; sl = 0.0; lb = 0
    for ht in halves:
        ent = lane_entries(chunk_masks(ht), mode)
        a, b = simulate(ent); it += a; sl += b
        lb += np.ceil(np.sum(ent, axis=0).max())
    print(f"{mode:10s}: {it / nsample:6.1f} iterations per half tile, lane slots busy {sl / (64 * it):.3f}   (max-lane bound: {lb / nsample:6.1f} iterations)")
