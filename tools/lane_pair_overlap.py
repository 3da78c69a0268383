#!/usr/bin/env python3
"""How many neighbours do two ADJACENT particles of the sorted order share?  (The lane-pair design of profiles/HISTORY.md §4.6 / VERDICT round 3,
next-5: two targets per lane, one gather and one bit walk per record of the UNION of their neighbour sets, two pair evaluations per
record.)  Adjacent = rows 2k, 2k+1 of the cell-sorted order.  Two orders: the lattice's own (file order inside a cell: the dam break at
rest) and a random order inside every cell (the developed flow: the in-cell order is the history of a few hundred stable sorts).
CPU only (scipy cKDTree).  usage: python tools/lane_pair_overlap.py [dp = 0.0085]"""
import os, sys
import numpy as np
from scipy.spatial import cKDTree
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0085
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
H = s.SimKernel.H
x = p.Position
cell = np.floor(np.abs(x) / H + 0.5).astype(np.int64) * np.sign(x).astype(np.int64)
key = (cell[:, 2] * 100000 + cell[:, 1]) * 100000 + cell[:, 0]
tree = cKDTree(x)
rng = np.random.default_rng(1)
for name, tie in (("lattice order inside a cell (at rest)", np.arange(len(x))), ("random order inside a cell (developed flow)", rng.permutation(len(x)))):
    order = np.lexsort((tie, key))
    xs = x[order]
    fluid = (p.Type[order] == 1)
    pairs = np.arange(0, len(xs) - 1, 2)
    pairs = pairs[fluid[pairs] & fluid[pairs + 1]]
    pick = rng.choice(pairs, size=min(4000, len(pairs)), replace=False)
    na = tree.query_ball_point(xs[pick], H)
    nb = tree.query_ball_point(xs[pick + 1], H)
    inter = np.array([len(set(a) & set(b)) for a, b in zip(na, nb)], float)
    la, lb = np.array([len(a) for a in na], float), np.array([len(b) for b in nb], float)
    union = la + lb - inter
    same_cell = (key[order][pick] == key[order][pick + 1]).mean()
    useful = (la + lb).sum() / union.sum()              # accepted pairs served per gathered record
    print(f"{name}: adjacent fluid pairs in the same cell {same_cell:.2f}; neighbours per target {la.mean():.0f}; union / single {union.mean() / la.mean():.2f}; "
          f"useful pairs per gathered record {useful:.2f} → gathers and bit walks per pair ×{1 / useful:.2f}, pair evaluations per useful pair ×{2 / useful:.2f}")
