"""TEST HARNESS — the Python twin of the slab driver that ships inside libsphmi.so (csrc/sphmi_multi.h).

The product path for more than one GPU is `sphmi_create` with a device list (one process, what the reference's Julia
caller needs) or `sphmi_create_rank` (one process per GPU, what bench.py uses): C++ host loop, RCCL linked into the
library.  This module drives the same kernels verb by verb (`sphmi_dd_*`) with `torch.distributed` and is kept because
its planning helpers (`SlabPlan`, `best_cuts`, `particle_work`, `choose_axis`) are the independent reference the C++
planner is tested against (tests/test_multi_gpu.py) and because its gloo path covers the message pattern on CPU.

Domain decomposition of the SPH hot path over the GPUs of one node.

The reference has no multi-process path at all; this module is new work for the MI355X
engine.  One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI), 1-D slabs cut on
cell-column boundaries so that every rank starts with (nearly) the same number of particles.  The slab
axis is the one whose columns split most evenly (``choose_axis``): a slab cut is one cell column coarse, so
the axis with many, evenly filled columns wins (y for the 3-D dam break: the water column occupies a short
stretch of x but the full width in y).

Why this shape
--------------
Interactions reach at most H and the cell size equals H, so a ONE-cell-column halo per side is enough
as long as owner and ghost copy use the same (stale) cell assignment — which they do, because particles
only change cells at a cell-list rebuild and the rebuild is a collective decision (the Δx criterion of
``src/SPHCellList.jl:744,758`` is evaluated on the global maxima).  A slab has at most two neighbours, each
on its own xGMI link, so the halo is two point-to-point messages per pass (no ring, no all-to-all); the
only collective in the step is one MAX-allreduce of four scalars that makes dt and the rebuild decision
bit-identical on all ranks.

Per step (mirrors ``Engine::step_once`` in csrc/sphmi_engine.hip):
    reductions → allreduce(MAX) → k_step_control on the device (Δx, dt, rebuild / stop flags; the host looks at
    them once per batch of up to 16 queued steps, sized to end at the step expected to ask for a rebuild) → [rebuild: migrate, re-ghost, sort] →
    halo(state A) ‖ predictor on interior tiles → predictor on slab-edge tiles →
    halo(half-step state H) ‖ corrector on interior tiles → corrector on slab-edge tiles
Ghost copies are ordinary entries of the rank's sorted particle array (type bits 0x80 / 0x40); the
kernels use them as neighbours and never compute, write or reduce them.  A tile (64 consecutive sorted
particles) is "slab-edge" when it holds an owned particle of the first / last cell column of the slab —
only those can see a ghost — so everything else runs while the two point-to-point messages are in flight.

Rebuild (collective): kill ghosts → send the particles whose cell column left the slab to the adjacent
rank → sort → send copies of the slab's first / last column as the neighbours' ghosts → sort again →
rebuild the halo index lists.  Both sorts are stable, so the k-th boundary particle of the sender is the
k-th ghost slot of the receiver and the per-step halo needs no indices on the wire.

The cuts move with the fluid: at a rebuild whose max/mean owned count exceeds 1.05 the ranks sum their column
histograms and re-cut (every cut stays between its old neighbours, so migration remains a neighbour exchange).

Moving bodies: ProgressMotion runs on owned particles and ghost copies alike before each halo pack (a prescribed
motion is the same function of time everywhere).  mDBC: the ghost layers are 2 + off columns wide (off = the largest
column distance between a boundary particle and its ghost node), every rank corrects the boundary particles it holds —
ghost copies included — after the halo of state A has landed, and pass 1 then runs without overlap.

Known limits: static slab axis.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from ._abi import SphmiConfig, SphmiProgress, make_config

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


class _Comm:
    """Point-to-point exchange with the two slab neighbours + the per-step MAX-allreduce.
    Device tensors go straight to RCCL; with the gloo backend (tests) they are staged through the host."""

    def __init__(self, rank: int, world: int, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.device = rank, world, device
        self.on_device = dist.get_backend() == "nccl"
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None

    def allreduce_max(self, values: np.ndarray) -> np.ndarray:
        t = self.torch.as_tensor(values, dtype=self.torch.float64)
        if self.on_device:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().numpy()

    def allreduce_i64(self, values: np.ndarray, op: str) -> np.ndarray:
        """Elementwise SUM / MIN / MAX of a small int64 host array over the ranks (rebuild-time bookkeeping)."""
        t = self.torch.as_tensor(np.ascontiguousarray(values, dtype=np.int64))
        if self.world > 1:
            if self.on_device:
                t = t.to(self.device)
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.cpu().numpy()

    def allreduce_max_bits(self, t):
        """In-place integer MAX over the ranks of a device int64 tensor (the engine's reduction bit patterns)."""
        if self.world > 1:
            if self.on_device:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            else:
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX)
                t.copy_(h)
        return t

    def start_exchange(self, send_left, send_right, recv_left_bytes: int, recv_right_bytes: int):
        """Post the two point-to-point exchanges; returns a token for ``finish_exchange``.  With RCCL the
        transfers run on the communicator's stream (ordered after the work already queued on the current
        stream), so kernels launched between start and finish overlap with them."""
        torch, dist = self.torch, self.dist
        stage = (lambda t: t) if self.on_device else (lambda t: t.cpu())
        dev = self.device if self.on_device else "cpu"
        ops, recv, keep = [], {}, []
        for peer, send, nrecv, key in ((self.left, send_left, recv_left_bytes, "L"),
                                       (self.right, send_right, recv_right_bytes, "R")):
            if peer is None:
                continue
            if nrecv > 0:
                recv[key] = torch.empty(nrecv, dtype=torch.uint8, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv[key], peer))
            if send is not None and send.numel() > 0:
                keep.append(stage(send).contiguous())
                ops.append(dist.P2POp(dist.isend, keep[-1], peer))
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, recv, keep

    def finish_exchange(self, token):
        works, recv, _keep = token
        for w in works:
            w.wait()            # RCCL: the current stream waits for the transfer; gloo: the host does
        out = []
        for key in ("L", "R"):
            t = recv.get(key)
            out.append(None if t is None else t.to(self.device))
        return out[0], out[1]

    def exchange(self, send_left, send_right, recv_left_bytes: int, recv_right_bytes: int):
        """send_*: uint8 device tensors (or None); returns the two received uint8 device tensors."""
        return self.finish_exchange(self.start_exchange(send_left, send_right, recv_left_bytes, recv_right_bytes))

    def exchange_counts(self, n_left: int, n_right: int) -> Tuple[int, int]:
        torch = self.torch
        dev = self.device if self.on_device else "cpu"
        mk = lambda v: torch.tensor([v], dtype=torch.int64, device=dev).view(torch.uint8)  # noqa: E731
        rl, rr = self.exchange(mk(n_left).to(self.device) if self.left is not None else None,
                               mk(n_right).to(self.device) if self.right is not None else None, 8, 8)
        g = lambda t: 0 if t is None else int(t.cpu().view(torch.int64)[0])  # noqa: E731
        return g(rl), g(rr)


class SphmiDdControl(C.Structure):
    _fields_ = [("steps_done", C.c_int64), ("total_time", C.c_double), ("last_dt", C.c_double), ("delta_x", C.c_double),
                ("need_rebuild", C.c_int32), ("stop", C.c_int32), ("error", C.c_int32), ("reserved", C.c_int32)]


class DistributedEngine:
    """Same ``advance`` / ``force_kernel_stats`` surface as ``engine.Engine``, on a slab of the domain."""

    def __init__(self, particles, setup, rank: int, world: int, local_device: int = 0,
                 device_float_bytes: int = 4, capacity_factor: float = 1.6, axis: Optional[int] = None,
                 overlap: bool = True, recut_imbalance: float = 1.05, plan: Optional[SlabPlan] = None):
        import torch
        from .engine import Engine, load_library
        self.torch = torch
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", local_device)
        torch.cuda.set_device(self.device)
        H_inv = setup.SimKernel.H_inv
        # initial ownership from the positions as the device will see them
        ft = np.float32 if device_float_bytes == 4 else np.float64
        D = particles.Position.shape[1]
        cols = [cell_x_of(particles.Position[:, a].astype(ft).astype(np.float64), H_inv) for a in range(D)]
        # halo width per axis: one column for the pair forces; with mDBC the ghost node of a boundary particle sits up
        # to `off` columns from it and is summed over its own 3 columns, and the ghost COPIES next to the slab are
        # corrected locally too (sphmi_dd_mdbc), so their ghost nodes' columns must be held as well: 2 + off
        from .config import SimpleMDBC
        self.mdbc = setup.SimMetaData.BMode is SimpleMDBC
        widths = [1] * D
        if self.mdbc:
            # `off` must not come out one short where a lattice sits exactly on cell edges and the device rounds the
            # hash the other way (fp32 arithmetic, FMA contraction): take the widest column distance any rounding
            # within ±1e-4 of a cell could give
            has = np.any(particles.GhostPoints != 0, axis=1)
            col = lambda u: (np.sign(u) * np.trunc(np.abs(u) + 0.5)).astype(np.int64)          # noqa: E731
            for a in range(D):
                ug = particles.GhostPoints[has, a].astype(ft).astype(np.float64) * H_inv
                ux = particles.Position[has, a].astype(ft).astype(np.float64) * H_inv
                d = 1e-4
                off = max(int(np.abs(col(ug + d) - col(ux - d)).max()), int(np.abs(col(ug - d) - col(ux + d)).max())) if has.any() else 0
                widths[a] = 2 + off
        work = particle_work(cols) if world > 1 else None
        self.axis = choose_axis(cols, world, [max(2, w) for w in widths], work) if axis is None else int(axis)
        self.halo_width = W = widths[self.axis]
        cx = cols[self.axis]
        self.plan = plan if plan is not None else SlabPlan.from_columns(cx, world, max(2, W), work)   # `plan`: start from given cuts
        self.plan.min_width = max(self.plan.min_width, W)
        mine = np.nonzero(self.plan.owner_of(cx) == rank)[0]
        self.n_total = len(particles)
        n_own = len(mine)
        # capacity: owned + the two ghost columns, with room for the fluid to pile up and for the cuts to move
        # (thin slabs of small cases carry ghost layers as large as the slab itself)
        lo_c, hi_c = int(cx.min()), int(cx.max())
        hist = np.bincount(cx - lo_c, minlength=hi_c - lo_c + 1)
        col = lambda c: int(hist[c - lo_c]) if lo_c <= c <= hi_c else 0          # noqa: E731
        s_lo, s_hi = max(self.plan.cx_lo[rank], lo_c), min(self.plan.cx_hi[rank], hi_c)
        ghosts = sum(col(s_lo - k) + col(s_hi + k) for k in range(1, W + 1))
        slack = col(s_lo - W - 1) + col(s_hi + W + 1) + 2 * int(hist.max())     # two more columns per side may arrive
        cap = int(capacity_factor * (n_own + ghosts)) + slack + 1024
        cap = max(cap, int(capacity_factor * (self.n_total / world)))
        cfg = make_config(cap, setup.SimConstants, setup.SimKernel, setup.SimMetaData, setup.SimViscosity,
                          setup.SimDensityDiffusion, device_float_bytes=device_float_bytes, host_float_bytes=8,
                          device=local_device)
        self.cfg = cfg
        self.eng = Engine(cfg)
        self.lib = load_library()
        self.h = self.eng._h
        self._declare()
        # everything of this engine — kernels, torch copies, RCCL transfers — is ordered on ONE non-default stream
        # (the legacy default stream synchronises implicitly with every blocking stream: ≈0.1 ms per step)
        self._main = torch.cuda.Stream(device=self.device)
        self._call("dd_set_stream", C.c_void_p(self._main.cuda_stream))
        INF = 1 << 30
        lo, hi = self.plan.cx_lo[rank], self.plan.cx_hi[rank]
        self._call("dd_set_slab", C.c_int(self.axis), C.c_int64(max(lo, -INF)), C.c_int64(min(hi, INF)),
                   C.c_int(rank > 0), C.c_int(rank < world - 1))
        self.overlap = overlap
        self.recut_imbalance = recut_imbalance          # re-cut the slabs at a rebuild when max/mean owned count exceeds this
        self.n_recuts = 0
        self._side = torch.cuda.Stream(device=self.device)
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        keep = [f(particles.Position[mine]), f(particles.Velocity[mine]), f(particles.Acceleration[mine]),
                f(particles.Density[mine]), np.ascontiguousarray(particles.Type[mine], dtype=np.uint8),
                np.ascontiguousarray(particles.ID[mine], dtype=np.int64),
                np.ascontiguousarray(particles.GroupMarker[mine], dtype=np.uint64)]
        gp = f(particles.GhostPoints[mine]) if self.mdbc else None
        order = np.ascontiguousarray(mine, dtype=np.int64)
        self._call("dd_upload", C.c_int64(n_own), *[a.ctypes.data_as(C.c_void_p) for a in keep],
                   gp.ctypes.data_as(C.c_void_p) if gp is not None else None, order.ctypes.data_as(C.c_void_p))
        self.moving = False
        self.comm = _Comm(rank, world, self.device)
        self.D = cfg.dims
        self.vbytes = 4 * cfg.device_float_bytes            # one V4 packet
        self.delta_x = 0.0
        self.total_time, self.iteration, self.last_dt = 0.0, 0, 0.0
        self.n_rebuilds = 0
        self._halo = None
        self._keep = [None, None]
        self._dx_rate = 0.0                     # Δx per step, from the last batch (sizes the next one)

    # -- ctypes plumbing -----------------------------------------------------------------------------
    def _declare(self):
        L = self.lib
        vp, i64, i32p = C.c_void_p, C.c_int64, C.c_void_p
        L.sphmi_dd_set_stream.argtypes = [vp, vp]
        L.sphmi_dd_upload.argtypes = [vp, i64] + [vp] * 9
        L.sphmi_dd_progress_motion.argtypes = [vp]
        L.sphmi_dd_mdbc.argtypes = [vp]
        L.sphmi_dd_count.argtypes = [vp, C.POINTER(i64)]
        L.sphmi_dd_cell_x.argtypes = [vp, vp]
        L.sphmi_dd_types.argtypes = [vp, vp]
        L.sphmi_dd_cell_x_dev.argtypes = [vp, vp]
        L.sphmi_dd_column_cost.argtypes = [vp, i64, C.c_int32, vp]
        L.sphmi_dd_types_dev.argtypes = [vp, vp]
        L.sphmi_dd_record_bytes.argtypes = [vp, i64, C.POINTER(i64)]
        L.sphmi_dd_gather.argtypes = [vp, i32p, i64, vp]
        L.sphmi_dd_kill.argtypes = [vp, i32p, i64]
        L.sphmi_dd_kill_ghosts.argtypes = [vp]
        L.sphmi_dd_append.argtypes = [vp, vp, i64, C.c_int]
        L.sphmi_dd_rebuild.argtypes = [vp]
        L.sphmi_dd_halo_pack.argtypes = [vp, C.c_int, i32p, i64, vp]
        L.sphmi_dd_halo_unpack.argtypes = [vp, C.c_int, i32p, i64, vp]
        L.sphmi_dd_reductions.argtypes = [vp, vp]
        L.sphmi_dd_reductions_dev.argtypes = [vp, vp]
        L.sphmi_dd_pass.argtypes = [vp, C.c_int, C.c_double]
        L.sphmi_dd_pass_part.argtypes = [vp, C.c_int, C.c_double, C.c_int]
        L.sphmi_dd_set_slab.argtypes = [vp, C.c_int, i64, i64, C.c_int, C.c_int]
        L.sphmi_dd_ctrl_init.argtypes = [vp, C.c_double, C.c_double, i64]
        L.sphmi_dd_step_control.argtypes = [vp, vp]
        L.sphmi_dd_ctrl_sync.argtypes = [vp, C.POINTER(SphmiDdControl)]
        L.sphmi_dd_ctrl_resume.argtypes = [vp]
        L.sphmi_dd_download_owned.argtypes = [vp, vp, vp, vp, vp, C.POINTER(i64)]
        L.sphmi_dd_progress.argtypes = [vp, C.POINTER(SphmiProgress)]

    def _call(self, name, *args):
        self.eng._check(getattr(self.lib, "sphmi_" + name)(self.h, *args))

    def set_motions(self, geometries):
        """MotionDetails of the Moving geometries (RunSimulation's MotionDefinition, src/SPHCellList.jl:846-850)."""
        self.eng.set_motions(geometries)
        self.moving = any(getattr(g, "Motion", None) is not None for g in geometries or ())

    def _count(self) -> int:
        n = C.c_int64()
        self._call("dd_count", C.byref(n))
        return n.value

    def _record_bytes(self, n: int) -> int:
        nb = C.c_int64()
        self._call("dd_record_bytes", C.c_int64(n), C.byref(nb))
        return nb.value

    # -- collective rebuild ------------------------------------------------------------------------
    def _cell_x_dev(self):
        """Global cell column of every live particle (ghost copies included) as a device tensor."""
        t = self.torch.empty(self._count(), dtype=self.torch.int32, device=self.device)
        self._call("dd_cell_x_dev", C.c_void_p(t.data_ptr()))
        return t

    def _types_dev(self):
        t = self.torch.empty(self._count(), dtype=self.torch.uint8, device=self.device)
        self._call("dd_types_dev", C.c_void_p(t.data_ptr()))
        return t

    def _where(self, mask):
        """Ascending int32 indices of the set entries — compaction on the device, nothing large crosses the bus."""
        return self.torch.nonzero(mask).flatten().to(self.torch.int32)

    def _gather_dev(self, idx):
        """Full records of the particles listed in the device index tensor → uint8 device tensor (None when empty)."""
        n = int(idx.numel())
        if n == 0:
            return None
        buf = self.torch.empty(self._record_bytes(n), dtype=self.torch.uint8, device=self.device)
        self._call("dd_gather", C.c_void_p(idx.data_ptr()), C.c_int64(n), C.c_void_p(buf.data_ptr()))
        return buf

    def _rebuild(self):
        """Everything that is per particle stays on the device (columns, type bits, index lists by compaction); the
        host sees counts and the handful of scalars of the load-balance decision."""
        torch = self.torch
        INF = 1 << 30
        cx = self._cell_x_dev()
        owned = (self._types_dev() & GHOST_MASK) == 0
        empty = torch.empty(0, dtype=torch.int32, device=self.device)
        # 0. load balance by WORK (candidates per particle, sphmi_dd_column_cost on the cell list of the previous rebuild):
        #    the rebuild is the only time particles change cells, so it is also when the cuts may move
        if self.world > 1 and self.recut_imbalance is not None and self.n_rebuilds > 0:
            co = cx[owned].to(torch.int64)
            ext = self.comm.allreduce_i64(torch.stack([-co.min(), co.max()]).cpu().numpy(), "MAX")
            gmin, gmax = -int(ext[0]), int(ext[1])
            cost = torch.zeros(gmax - gmin + 1, dtype=torch.int64, device=self.device)
            self._call("dd_column_cost", C.c_int64(gmin), C.c_int32(gmax - gmin + 1), C.c_void_p(cost.data_ptr()))
            mine = int(cost.sum())
            wmax = int(self.comm.allreduce_i64(np.array([mine]), "MAX")[0])
            total = int(self.comm.allreduce_i64(np.array([mine]), "SUM")[0])
            if wmax * self.world > self.recut_imbalance * total:
                hist = self.comm.allreduce_i64(cost.cpu().numpy(), "SUM")
                plan = self.plan.recut(gmin, hist)
                if plan.cuts() != self.plan.cuts():
                    self.plan = plan
                    self._call("dd_set_slab", C.c_int(self.axis), C.c_int64(max(plan.cx_lo[self.rank], -INF)),
                               C.c_int64(min(plan.cx_hi[self.rank], INF)), C.c_int(self.rank > 0), C.c_int(self.rank < self.world - 1))
                    self.n_recuts += 1
        lo, hi = self.plan.cx_lo[self.rank], self.plan.cx_hi[self.rank]
        # 1. ghosts die, leavers migrate to the adjacent rank
        go_l = self._where(owned & (cx < lo)) if self.comm.left is not None else empty
        go_r = self._where(owned & (cx > hi)) if self.comm.right is not None else empty
        skipped = False
        if go_l.numel():
            skipped |= bool((cx[go_l.long()] < self.plan.cx_lo[self.rank - 1]).any())
        if go_r.numel():
            skipped |= bool((cx[go_r.long()] > self.plan.cx_hi[self.rank + 1]).any())
        if skipped:
            raise RuntimeError("domain decomposition: a particle skipped a whole slab between two rebuilds")
        self._call("dd_kill_ghosts")
        sl, sr = self._gather_dev(go_l), self._gather_dev(go_r)
        nl, nr = self.comm.exchange_counts(int(go_l.numel()), int(go_r.numel()))
        rl, rr = self.comm.exchange(sl, sr, self._record_bytes(nl) if nl else 0, self._record_bytes(nr) if nr else 0)
        leavers = torch.cat([go_l, go_r])
        if leavers.numel():
            self._call("dd_kill", C.c_void_p(leavers.data_ptr()), C.c_int64(int(leavers.numel())))
        for buf, n in ((rl, nl), (rr, nr)):
            if n:
                self._call("dd_append", C.c_void_p(buf.data_ptr()), C.c_int64(n), C.c_int(0))
        torch.cuda.current_stream(self.device).synchronize()
        self._call("dd_rebuild")
        # 2. the first / last column(s) of the slab become the neighbours' ghost layer
        W = self.halo_width
        cx = self._cell_x_dev()
        b_l = self._where(cx < lo + W) if self.comm.left is not None else empty
        b_r = self._where(cx > hi - W) if self.comm.right is not None else empty
        sl, sr = self._gather_dev(b_l), self._gather_dev(b_r)
        n_bl, n_br = int(b_l.numel()), int(b_r.numel())
        nl, nr = self.comm.exchange_counts(n_bl, n_br)
        rl, rr = self.comm.exchange(sl, sr, self._record_bytes(nl) if nl else 0, self._record_bytes(nr) if nr else 0)
        if nl:
            self._call("dd_append", C.c_void_p(rl.data_ptr()), C.c_int64(nl), C.c_int(GHOST_LEFT))
        if nr:
            self._call("dd_append", C.c_void_p(rr.data_ptr()), C.c_int64(nr), C.c_int(GHOST_RIGHT))
        torch.cuda.current_stream(self.device).synchronize()
        self._call("dd_rebuild")
        # 3. halo index lists in the final order (stable sorts ⇒ k-th sender entry ↔ k-th ghost slot; the same holds
        #    for the sub-lists of ONE column, because both sides see the same positions).  State A travels with all
        #    `halo_width` columns (mDBC reads them), the half-step state H with the one column the pair forces reach.
        cx = self._cell_x_dev()
        ty = self._types_dev()
        owned = (ty & GHOST_MASK) == 0
        g_l, g_r = (ty & GHOST_LEFT) != 0, (ty & GHOST_RIGHT) != 0
        vb = 2 * self.vbytes
        mk = lambda n: torch.empty(max(n, 1) * vb, dtype=torch.uint8, device=self.device)  # noqa: E731
        self._halo = []
        for w in (W, 1):
            send_l = self._where(owned & (cx < lo + w)) if self.comm.left is not None else empty
            send_r = self._where(owned & (cx > hi - w)) if self.comm.right is not None else empty
            slot_l = self._where(g_l & (cx >= lo - w))
            slot_r = self._where(g_r & (cx <= hi + w))
            if w == W:
                assert send_l.numel() == n_bl and send_r.numel() == n_br, "boundary columns changed between the two sorts"
                assert slot_l.numel() == nl and slot_r.numel() == nr
            self._halo.append(dict(send_l=send_l, send_r=send_r, slot_l=slot_l, slot_r=slot_r,
                                   n_send_l=int(send_l.numel()), n_send_r=int(send_r.numel()),
                                   n_slot_l=int(slot_l.numel()), n_slot_r=int(slot_r.numel()),
                                   buf_l=mk(int(send_l.numel())), buf_r=mk(int(send_r.numel()))))
            if W == 1:
                self._halo.append(self._halo[0])
                break
        self.n_rebuilds += 1

    def _halo_start(self, which: int):
        """Pack the slab-edge columns of state set `which` (0 = A, 1 = H) and post the exchange."""
        hl, p = self._halo[which], C.c_void_p
        vb = 2 * self.vbytes
        if hl["n_send_l"]:
            self._call("dd_halo_pack", C.c_int(which), p(hl["send_l"].data_ptr()), C.c_int64(hl["n_send_l"]), p(hl["buf_l"].data_ptr()))
        if hl["n_send_r"]:
            self._call("dd_halo_pack", C.c_int(which), p(hl["send_r"].data_ptr()), C.c_int64(hl["n_send_r"]), p(hl["buf_r"].data_ptr()))
        return self.comm.start_exchange(hl["buf_l"][:hl["n_send_l"] * vb] if hl["n_send_l"] else None,
                                        hl["buf_r"][:hl["n_send_r"] * vb] if hl["n_send_r"] else None,
                                        hl["n_slot_l"] * vb, hl["n_slot_r"] * vb)

    def _halo_finish(self, which: int, token):
        """Wait for the exchange and refresh the ghost copies of state set `which`."""
        hl, p = self._halo[which], C.c_void_p
        rl, rr = self.comm.finish_exchange(token)
        if hl["n_slot_l"]:
            self._call("dd_halo_unpack", C.c_int(which), p(hl["slot_l"].data_ptr()), C.c_int64(hl["n_slot_l"]), p(rl.data_ptr()))
        if hl["n_slot_r"]:
            self._call("dd_halo_unpack", C.c_int(which), p(hl["slot_r"].data_ptr()), C.c_int64(hl["n_slot_r"]), p(rr.data_ptr()))
        self._keep[which] = (rl, rr, token)   # receive / send buffers stay alive until this set is exchanged again

    def _pass(self, which: int, dt: float):
        """One neighbour pass with its halo.  Interior tiles run on the main stream while the messages are in
        flight; the unpack and the slab-edge tiles go to a SIDE stream that only waits for the messages, so the
        edge tiles start as soon as the halo has landed and share the chip with the interior launch instead of
        forming a second, poorly filled launch behind it."""
        torch = self.torch
        if self.moving:
            self._call("dd_progress_motion")         # :765 / :787 — owned and ghost copies move alike, then the pack
        token = self._halo_start(which - 1)
        if not self.overlap or (self.mdbc and which == 1):
            # mDBC (:772) reads the fluid of state A in the ghost layers and rewrites the boundary densities that every
            # tile of pass 1 may read: halo → mDBC → the whole pass, nothing to overlap
            self._halo_finish(which - 1, token)
            if self.mdbc and which == 1:
                self._call("dd_mdbc")
            self._call("dd_pass", C.c_int(which), C.c_double(dt))
            return
        main = torch.cuda.current_stream(self.device)
        side = self._side
        side.wait_stream(main)                       # everything queued so far: the previous pass, the pack
        self._call("dd_pass_part", C.c_int(which), C.c_double(dt), C.c_int(1))
        with torch.cuda.stream(side):
            self._call("dd_set_stream", C.c_void_p(side.cuda_stream))
            try:
                self._halo_finish(which - 1, token)  # the side stream waits for the transfers, then unpacks
                self._call("dd_pass_part", C.c_int(which), C.c_double(dt), C.c_int(2))
            finally:
                self._call("dd_set_stream", C.c_void_p(main.cuda_stream))
        main.wait_stream(side)                       # the next pack / the reductions need the edge tiles

    # -- the SimulationLoop of src/SPHCellList.jl:727-805, distributed --------------------------------
    BATCH = 16     # most steps queued between two looks at the control flags

    def advance(self, t_target: float, max_steps: int = -1) -> SphmiProgress:
        with self.torch.cuda.stream(self._main):
            return self._advance(t_target, max_steps)

    def _advance(self, t_target: float, max_steps: int = -1) -> SphmiProgress:
        """Every per-step decision (Δx, Δt, loop bound, rebuild criterion) is taken on the device by the engine's
        k_step_control from the MAX-allreduced reduction slots — identical on every rank — so the host queues BATCH
        steps (reductions → allreduce → control → two passes with their halos) and synchronises once per batch; the
        kernels of a step the control cancelled return at once, the exchanges still match on both sides."""
        cfg = self.cfg
        self.delta_x = 1.0 + cfg.h                                   # :739
        steps = 0
        # persistent: sphmi_dd_reductions_dev MAX-merges into it and k_step_control zeroes it when a step consumes it, so the
        # maxima a control left unused at the end of an interval are still there for the first Δt of the next one
        if getattr(self, "_red_t", None) is None:
            self._red_t = self.torch.zeros(4, dtype=self.torch.int64, device=self.device)
        red_t = self._red_t
        self._call("dd_ctrl_init", C.c_double(self.delta_x), C.c_double(t_target), C.c_int64(max_steps))
        st = SphmiDdControl()
        first = True
        while True:
            # A step that asks for a rebuild cancels the rest of its batch, and a cancelled step still pays its allreduce
            # and its four point-to-point messages: queue up to the step that is EXPECTED to ask — Δx grows by 4·max|Δx|
            # a step, at a rate that changes slowly — and no further.  Every rank computes the same number from the same
            # (allreduced) values.
            batch = self.BATCH
            if self._dx_rate > 0.0:
                batch = max(1, min(batch, int((cfg.h - self.delta_x) / self._dx_rate) + 1))
            if max_steps >= 0:
                batch = max(1, min(batch, max_steps - steps))
            if first:
                batch = 1          # the loop re-arms Δx = 1 + h: the first control of a call always asks for a rebuild
                first = False
            dx0, steps0 = self.delta_x, steps
            for _ in range(batch):
                # local maxima → global maxima → decisions, without leaving the device
                self._call("dd_reductions_dev", C.c_void_p(red_t.data_ptr()))
                self.comm.allreduce_max_bits(red_t)
                self._call("dd_step_control", C.c_void_p(red_t.data_ptr()))
                if self._halo is not None:          # before the first rebuild there is no ghost layer to exchange
                    self._pass(1, 0.0)
                    self._pass(2, 0.0)
            self._call("dd_ctrl_sync", C.byref(st))                   # the one host synchronisation of the batch
            steps = st.steps_done
            self.total_time, self.last_dt, self.delta_x = st.total_time, st.last_dt, st.delta_x
            grown = (steps - steps0) + (1 if st.need_rebuild else 0)          # controls that added their 4·max|Δx|
            if grown > 0 and dx0 < cfg.h and st.delta_x > dx0:
                self._dx_rate = (st.delta_x - dx0) / grown
            if st.error == 2:
                raise RuntimeError("non-positive density produced on some rank")
            if st.error:
                raise RuntimeError(f"non-positive or NaN dt at iteration {self.iteration + steps}")
            if st.need_rebuild:
                self._rebuild()
                self.delta_x = 0.0
                self._call("dd_ctrl_resume")
                continue
            if st.stop or not (self.total_time <= t_target) or (0 <= max_steps <= steps):
                break
        self.iteration += steps
        self.torch.cuda.current_stream(self.device).synchronize()
        prog = SphmiProgress()
        self._call("dd_progress", C.byref(prog))
        prog.iteration, prog.steps_done, prog.n_rebuilds = self.iteration, steps, self.n_rebuilds
        prog.total_time, prog.last_dt, prog.delta_x = self.total_time, self.last_dt, self.delta_x
        return prog

    def force_kernel_stats(self, reset: bool = False):
        return self.eng.force_kernel_stats(reset)

    def download_owned(self) -> dict:
        self._main.synchronize()
        n = self._count()
        pos = np.empty((n, self.D)); vel = np.empty((n, self.D)); rho = np.empty(n); ids = np.empty(n, dtype=np.int64)
        m = C.c_int64()
        self._call("dd_download_owned", pos.ctypes.data_as(C.c_void_p), vel.ctypes.data_as(C.c_void_p),
                   rho.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.byref(m))
        k = m.value
        return {"Position": pos[:k], "Velocity": vel[:k], "Density": rho[:k], "ID": ids[:k]}

    def gather_all(self) -> Optional[dict]:
        """Owned particles of every rank, concatenated on rank 0 (tests / output)."""
        import torch.distributed as dist
        parts: List[Optional[dict]] = [None] * self.world if self.rank == 0 else None
        dist.gather_object(self.download_owned(), parts, dst=0)
        if self.rank != 0:
            return None
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def make_distributed_engine(dp: float, setup, rank: int, world: int, local_device: int):
    """bench.py hook: every rank generates the (deterministic) lattice and keeps its slab."""
    from .cases import dam_break_3d
    particles = dam_break_3d(dp)
    return DistributedEngine(particles, setup, rank, world, local_device), len(particles)
