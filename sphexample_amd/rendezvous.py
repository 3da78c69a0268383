"""Launcher-side glue of a one-process-per-GPU run (`torchrun bench.py --gpus N`, sphmi_create_rank).

torch.distributed is the RENDEZVOUS only — gloo over TCP: the 128-byte RCCL id travels from rank 0 to the others, the
timed region is bracketed by barriers, the wall time is the max over ranks, and every rank learns whether ALL ranks got
their slab engine.  Halos, migration and the per-step MAX-allreduce run inside libsphmi.so (csrc/sphmi_multi.h) on RCCL.
The reference has no counterpart (one Julia process, src/SPHCellList.jl:883); this is what a multi-process host driver
needs around the C ABI.  Works without a GPU (tests/test_distributed.py runs it with 2 and 3 CPU processes).
"""
from __future__ import annotations

import os
import socket
from typing import Optional


def free_port() -> int:
    """A TCP port nobody listens on right now (single-node launches without a launcher-provided MASTER_PORT)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StdoutToStderr:
    """gloo announces its connections on the C++ side's stdout ("[Gloo] Rank 0 is connected to …"); bench.py's stdout is ONE JSON
    line.  File descriptor 1 points at stderr while the process group is set up."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Rendezvous:
    def __init__(self, rank: int, world: int, master_addr: Optional[str] = None, master_port: Optional[int] = None):
        import torch.distributed as dist
        self.rank, self.world, self._dist = rank, world, dist
        self._own = not dist.is_initialized()
        if not self._own:
            return
        with _StdoutToStderr():
            self._init(rank, world, master_addr, master_port)

    def _init(self, rank, world, master_addr, master_port):
        dist = self._dist
        if master_port is None and os.environ.get("MASTER_PORT"):
            # under a launcher (torch.distributed.run exports MASTER_ADDR / MASTER_PORT and may host the store itself):
            # the launcher's own rendezvous, untouched
            os.environ.setdefault("MASTER_ADDR", master_addr or "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return
        if master_port is None:
            if world > 1:
                raise RuntimeError("MASTER_PORT is not set: launch with `python -m torch.distributed.run --master-addr 127.0.0.1 "
                                   "--master-port P …` (every rank must name the same port)")
            master_port = free_port()                # a one-rank world: nobody else has to know the port
        addr = master_addr or os.environ.get("MASTER_ADDR") or "127.0.0.1"
        dist.init_process_group("gloo", init_method=f"tcp://{addr}:{int(master_port)}", rank=rank, world_size=world)

    def broadcast_bytes(self, payload: Optional[bytes], src: int = 0) -> bytes:
        box = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self) -> None:
        self._dist.barrier()

    def max(self, x: float) -> float:
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok: bool) -> bool:
        """True on every rank iff every rank passed True."""
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def gather_strings(self, text: str):
        """Every rank's string, on every rank (error texts of a failed set-up: the line names rank, peer and call)."""
        out = [None] * self.world
        self._dist.all_gather_object(out, text)
        return out

    def close(self, after_error: bool = False) -> None:
        """`after_error`: a peer may be gone already — no barrier (it would wait for ever), just leave the group."""
        if self._own and self._dist.is_initialized():
            try:
                if not after_error:
                    self._dist.barrier()
            finally:
                self._dist.destroy_process_group()


def library_check(shm: bool) -> None:
    """The default local check of create_rank_engine: libsphmi.so loads (and is built first when stale), RCCL binds."""
    from .engine import load_library, rccl_probe
    load_library()
    if not shm:
        rccl_probe()                                  # dlopen + symbols only — no ncclGetUniqueId: that starts a bootstrap root per call


def create_rank_engine(rdv: Rendezvous, make, transport_env: str = "SPHMI_TRANSPORT", preflight=None, timeout: float = None,
                       local_check=library_check, make_id=None):
    """One slab engine per rank with the id handed round, COLLECTIVELY: every rank learns whether all ranks succeeded.
    `make(unique_id)` builds this rank's engine (sphexample_amd.engine.make_engine(..., rank=, world=, unique_id=)).
    Returns (engine | None, list of the ranks' error texts).  The id is RCCL's (sphmi_rccl_unique_id, asked for ONCE, on rank 0)
    unless the shared-memory transport is selected in the environment, where any 128 random bytes do; `make_id` overrides both
    (a caller with its own engine factory).

    Two phases (round-3 advice: a rank that failed BEFORE ncclCommInitRank left its peers waiting inside it for ever):
      1. local — everything that can fail without a peer: `local_check(shm)` (default: the library loads, RCCL binds — without
         making an id; None: nothing, for callers whose `make` does not go through libsphmi) and `preflight()` of the caller
         (device present, memory for the slab …).  All ranks agree on the outcome (all_ok) BEFORE anyone enters the communicator set-up.
      2. collective — `make(uid)`.  A communicator set-up that cannot reach a peer does not fail, it waits: `timeout` seconds
         (or $SPHMI_SETUP_TIMEOUT) arm a watchdog that ends the process with a message and exit code 3."""
    shm = os.environ.get(transport_env) == "shm"
    err = ""
    try:
        if local_check is not None:
            local_check(shm)
        if preflight is not None:
            preflight()
    except Exception as exc:                          # noqa: BLE001
        err = f"rank {rdv.rank} (local set-up): {exc}"
    texts = rdv.gather_strings(err)
    if not rdv.all_ok(not err):
        return None, [t for t in texts if t]
    uid = None
    if rdv.rank == 0:
        try:
            if make_id is not None:
                uid = make_id()
            elif shm:
                uid = os.urandom(128)
            else:
                from .engine import rccl_unique_id
                uid = rccl_unique_id()
        except Exception as exc:                      # noqa: BLE001 — every rank must learn it, not hang
            err = f"rank 0: sphmi_rccl_unique_id: {exc}"
    uid = rdv.broadcast_bytes(uid, src=0)
    eng = None
    if uid is not None:
        watchdog = None
        limit = timeout if timeout is not None else float(os.environ.get("SPHMI_SETUP_TIMEOUT", "0") or 0)
        if limit > 0:
            import sys
            import threading

            def give_up():
                print(f"[sphmi] rank {rdv.rank}: the slab engine was not set up within {limit:.0f} s (a peer that never reached "
                      f"ncclCommInitRank?)", file=sys.stderr, flush=True)
                os._exit(3)
            watchdog = threading.Timer(limit, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            eng = make(uid)
        except Exception as exc:                      # noqa: BLE001
            err = f"rank {rdv.rank}: {exc}"
        if watchdog is not None:
            watchdog.cancel()
    texts = rdv.gather_strings(err)
    if not rdv.all_ok(eng is not None):
        if eng is not None:
            eng.close()
        return None, [t for t in texts if t]
    return eng, []


__all__ = ["Rendezvous", "create_rank_engine", "free_port"]
