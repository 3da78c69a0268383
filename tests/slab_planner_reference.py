"""The slab PLANNER in Python — the independent reference the C++ planner of libsphmi.so is tested against.
TEST INFRASTRUCTURE (it lived in the product package as sphexample_amd/distributed.py until round 6; nothing in the package used it).

The product path for more than one GPU lives inside the library (csrc/sphmi_multi.h): `sphmi_create` with a device list
(one process, what the reference's Julia caller needs) or `sphmi_create_rank` (one process per GPU, what bench.py uses).
This module holds only the host-side planning arithmetic, written separately in numpy: which axis to cut (`choose_axis`),
where (`best_cuts`, `SlabPlan`: exact lightest-heaviest-slab partition of the column work histogram; `recut` keeps every
cut between its old neighbours so that migration stays a neighbour exchange), the work measure itself (`particle_work`:
candidates in the 3^D cells around a particle, the measure of the kernel's tile schedule) and the per-step Δt / Δx
arithmetic (`step_control`, src/TimeStepping.jl:30-43).  tests/test_multi_gpu.py and tests/test_distributed.py compare
`sphmi_plan_slabs` / `sphmi_multi_column_cost` with it.  (Rounds 1-2 also kept a Python twin of the slab DRIVER here, over
`sphmi_dd_*` verbs of the C ABI; the verbs and the twin are gone — one driver, one public header.)

The reference has no multi-process path.  Interactions reach at most H and the cell size equals H, so a
ONE-cell-column halo per side is enough as long as owner and ghost copy use the same (stale) cell assignment — which they
do, because particles only change cells at a cell-list rebuild and the rebuild is a collective decision (the Δx criterion
of ``src/SPHCellList.jl:744,758`` is evaluated on the global maxima).  A slab has at most two neighbours, each on its own
xGMI link, so the halo is two point-to-point messages per pass (no ring, no all-to-all); the only collective in the step
is one MAX-allreduce of four scalars that makes dt and the rebuild decision bit-identical on all ranks.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from sphexample_amd._abi import SphmiConfig

GHOST_LEFT, GHOST_RIGHT, GHOST_MASK = 0x80, 0x40, 0xC0


def cell_x_of(x: np.ndarray, H_inv: float) -> np.ndarray:
    """map_floor of src/SPHCellList.jl:56-61 on one coordinate (round half away from zero)."""
    return (np.sign(x) * np.trunc(np.abs(x) * H_inv + 0.5)).astype(np.int64)


def best_cuts(hist: np.ndarray, world: int, width: int, lo_b=None, hi_b=None) -> Optional[List[int]]:
    """Cut positions c_1 < … < c_{world-1} (slab r = columns c_r … c_{r+1}-1, c_0 = 0, c_world = len(hist)) that
    MINIMISE THE HEAVIEST SLAB, every slab at least `width` columns, cut r within [lo_b[r], hi_b[r]] when given.
    Exact (dynamic programme over the columns): with a dozen columns per slab one column is a tenth of a slab's work,
    and cutting at the quantiles of the cumulative work can be that far from the best partition.  None: infeasible."""
    n = len(hist)
    cum = np.concatenate([[0.0], np.cumsum(hist, dtype=np.float64)])
    INF = float("inf")
    f = np.full(n + 1, INF); f[0] = 0.0                       # f[c]: best heaviest-slab value with the current cut at c
    back = []
    for r in range(1, world + 1):
        g = np.full(n + 1, INF); arg = np.zeros(n + 1, dtype=np.int64)
        cs = [n] if r == world else range(max(r * width, 0 if lo_b is None else lo_b[r]),
                                          min(n - (world - r) * width, n if hi_b is None else hi_b[r]) + 1)
        for c in cs:
            prev = np.arange(0, c - width + 1)
            if len(prev) == 0:
                continue
            v = np.maximum(f[prev], cum[c] - cum[prev])
            k = int(np.argmin(v))
            g[c], arg[c] = v[k], prev[k]
        f = g; back.append(arg)
    if not np.isfinite(f[n]):
        return None
    cuts, c = [], n
    for r in range(world, 0, -1):
        c = int(back[r - 1][c]); cuts.append(c)
    return list(reversed(cuts))[1:]                           # drop c_0 = 0


def particle_work(cols: List[np.ndarray]) -> np.ndarray:
    """What a particle costs in the neighbour kernel: the number of candidates in the 3^D cells around its own (the
    measure of the engine's tile schedule and of sphmi_dd_column_cost).  A dry wall particle has a tenth of an interior
    fluid particle's, so cuts that equalise particle COUNTS leave the rank with the dry walls short of work."""
    lo = [int(c.min()) for c in cols]
    dims = [int(c.max()) - l + 3 for c, l in zip(cols, lo)]                    # one cell of padding per side
    lin = np.zeros(len(cols[0]), dtype=np.int64)
    for c, l, d in zip(reversed(cols), reversed(lo), reversed(dims)):           # last axis slowest, like the cell sort
        lin = lin * d + (c - l + 1)
    grid = np.bincount(lin, minlength=int(np.prod(dims))).reshape(list(reversed(dims)))
    box = grid.astype(np.int64)
    for ax in range(box.ndim):                                                  # separable 3-wide box sum
        up, dn = np.roll(box, 1, axis=ax), np.roll(box, -1, axis=ax)            # padding cells are empty: no wrap-around
        box = box + up + dn
    return box.reshape(-1)[lin]


def choose_axis(cols: List[np.ndarray], world: int, min_width=2, weights: Optional[np.ndarray] = None) -> int:
    """Slab axis: the one whose best column cuts leave the lightest heaviest rank (by `weights` = particle_work, else by
    count).  Axes within 1 % of the best are a tie, won by the thinnest ghost layers (fewest particles in the columns next
    to the cuts: fewer slab-edge tiles waiting for the halo), then by the slowest sort axis (its ghost layers are
    contiguous runs of the cell-sorted arrays).  `min_width`: columns a slab must keep — a number, or one per axis (the
    halo width, which depends on the axis with mDBC)."""
    cand = []
    for ax, cx in enumerate(cols):
        try:
            plan = SlabPlan.from_columns(cx, world, min_width if np.isscalar(min_width) else min_width[ax], weights)
        except ValueError:
            continue
        load = np.bincount(plan.owner_of(cx), weights=weights, minlength=world).max()
        edge = np.zeros(0, dtype=np.int64) if world == 1 else np.concatenate([[c - 1, c] for c in plan.cuts()])
        cand.append((float(load), int(np.isin(cx, edge).sum()), ax))
    if not cand:
        raise ValueError("no axis has enough cell columns per rank: too many ranks for this domain")
    best = min(c[0] for c in cand)
    tie = [c for c in cand if c[0] <= 1.01 * best]
    return min(tie, key=lambda c: (c[1], -c[2]))[2]


@dataclass
class SlabPlan:
    """Static slab cuts along one axis: rank r owns the cell columns cx_lo[r] … cx_hi[r] (inclusive)."""
    cx_lo: List[int]
    cx_hi: List[int]
    min_width: int = 2          # columns per slab: at least the halo width (a ghost layer comes from ONE neighbour)

    @property
    def world(self) -> int:
        return len(self.cx_lo)

    @staticmethod
    def from_columns(cx: np.ndarray, world: int, min_width: int = 2, weights: Optional[np.ndarray] = None) -> "SlabPlan":
        """Equal-WORK cuts on column boundaries (`weights`: particle_work; equal particle counts without) — a uniform
        spatial cut would put the whole initial water column on a quarter of the ranks."""
        lo, hi = int(cx.min()), int(cx.max())
        hist = np.bincount(cx - lo, weights=weights, minlength=hi - lo + 1)
        inner = best_cuts(hist, world, min_width if world > 1 else 1)
        if inner is None:
            raise ValueError(f"a slab would be narrower than {min_width} cell columns: too many ranks for this domain")
        cuts = [lo] + [lo + c for c in inner] + [hi + 1]
        INF = 1 << 30
        cx_lo = [(-INF if r == 0 else cuts[r]) for r in range(world)]
        cx_hi = [(INF if r == world - 1 else cuts[r + 1] - 1) for r in range(world)]
        return SlabPlan(cx_lo, cx_hi, min_width)

    def cuts(self) -> List[int]:
        """Interior cut positions: cut r (1 ≤ r < world) is the first column of rank r."""
        return [self.cx_lo[r] for r in range(1, self.world)]

    def recut(self, col0: int, hist: np.ndarray) -> "SlabPlan":
        """Best cuts (lightest heaviest slab) for the CURRENT global column histogram (`hist[k]` = work, or particles,
        of column col0 + k), with every cut kept between its two old neighbours (so a particle changes rank by at most
        one — migration stays a neighbour exchange) and every slab at least `min_width` columns wide."""
        world, w = self.world, self.min_width
        n = len(hist)
        old = [col0] + self.cuts() + [col0 + n]
        # cut r may move between its old neighbours, a slab width away from both
        lo_b = [0] + [old[r - 1] + w - col0 for r in range(1, world)] + [n]
        hi_b = [0] + [old[r + 1] - w - col0 for r in range(1, world)] + [n]
        inner = best_cuts(np.asarray(hist, dtype=np.float64), world, w, lo_b, hi_b)
        new = old[:-1] if inner is None else [old[0]] + [col0 + c for c in inner]
        INF = 1 << 30
        return SlabPlan([(-INF if r == 0 else new[r]) for r in range(world)],
                        [(INF if r == world - 1 else new[r + 1] - 1) for r in range(world)], w)

    def owner_of(self, cx: np.ndarray) -> np.ndarray:
        bounds = np.array([self.cx_lo[r] for r in range(1, self.world)], dtype=np.int64)
        return np.searchsorted(bounds, cx, side="right")


def step_control(red: np.ndarray, delta_x: float, cfg: SphmiConfig) -> Tuple[float, float, bool]:
    """Δx accumulation, Δt and the rebuild decision from the GLOBAL reductions
    (src/SPHCellList.jl:706-724,744,758; src/TimeStepping.jl:30-43) — same arithmetic as step_once."""
    maxdisp = float(np.sqrt(red[0]))
    visc = float(red[1])
    amax = float(np.sqrt(red[2]))
    delta_x = delta_x + 4.0 * maxdisp
    with np.errstate(divide="ignore"):
        dt1 = float(np.sqrt(np.float64(cfg.h) / np.float64(amax))) if amax > 0 else float("inf")
    dt2 = cfg.h / (cfg.c0 + visc)
    dt = cfg.CFL * min(dt1, dt2)
    return delta_x, dt, delta_x >= cfg.h
