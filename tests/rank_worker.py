"""One process of a rank-mode run (sphmi_create_rank): spawned `world` times by tests/test_rank_mode.py.

  python rank_worker.py selftest <unique id, hex> <rank> <world> <bytes per message>
  python rank_worker.py run <unique id, hex> <rank> <world> <case> <steps> <float bytes> <out dir> <advance calls> <slab axis | -1>
  python rank_worker.py run_slabs <unused> 0 <world> <case> …          (one process holding all `world` slabs on GPU 0)

`run` creates slab `rank` of the case on GPU 0 with SPHMI_TRANSPORT=shm (the environment of the spawner) — or through the RCCL branch
with the checking double behind $SPHMI_RCCL_LIB (tests/test_mock_rccl_gpu.py), or, with $SPHMI_TEST_DEVICE_PER_RANK=1
(tests/test_multi_device_gpu.py on a box with several GPUs), on GPU `rank` over the real RCCL —
advances it and stores what this process owns plus the loop counters; the spawner compares the union with a one-device handle.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    mode, uid, rank, world = sys.argv[1], bytes.fromhex(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from sphexample_amd.engine import load_library
    lib = load_library(rebuild_if_stale=False)
    if mode == "selftest":
        lib.sphmi_shm_selftest.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int64]
        rc = lib.sphmi_shm_selftest(uid, rank, world, int(sys.argv[5]))
        if rc:
            lib.sphmi_last_error.restype = C.c_char_p
            print(f"rank {rank}: rc {rc}: {lib.sphmi_last_error(None).decode()}", file=sys.stderr)
        sys.exit(rc)
    case, steps, fb, out_dir, calls, axis = sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), sys.argv[8], int(sys.argv[9]), int(sys.argv[10])
    import conftest
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + case)()
    device = rank if os.environ.get("SPHMI_TEST_DEVICE_PER_RANK") == "1" else 0
    if mode == "run_slabs":        # ONE process, `world` slabs sharing GPU 0 (with SPHMI_TRANSPORT=rccl + a substitute library: the ncclCommInitAll branch)
        eng = make_engine(p, s, device_float_bytes=fb, devices=[0] * world, slab_axis=None if axis < 0 else axis)
    else:
        eng = make_engine(p, s, device_float_bytes=fb, device=device, rank=rank, world=world, unique_id=uid,
                          slab_axis=None if axis < 0 else axis)
    prog = []
    if os.environ.get("SPHMI_TEST_REUPLOAD"):
        # an ODD number of steps, then the same particle set again: the step parity restarts, the mailbox sequence does not (round-5 advice)
        eng.advance(1e9, max_steps=int(os.environ["SPHMI_TEST_REUPLOAD"]))
        eng.upload_particles(p)
    for _ in range(calls):
        pr = eng.advance(1e9, max_steps=steps // calls)
        prog.append([pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter, pr.total_time, pr.last_dt])
    info = eng.multi_info()
    info = np.array([info.world, info.n_local, info.transport, info.axis, info.halo_width, info.n_recuts, info.reserved])
    d = eng.download(("Position", "Density", "ID", "Velocity"))
    if os.environ.get("SPHMI_TEST_FORCES_ONCE") == "1":        # the parity hook on slabs: one more collective rebuild + a forces-only pass
        d["drhodt"], d["acc"] = eng.forces_once()
        n_own = eng.owned_count() if mode == "run" else len(d["drhodt"])
        d["drhodt"], d["acc"] = d["drhodt"][:n_own], d["acc"][:n_own]
        d["ID_forces"] = eng.download(("ID",))["ID"]
    mock = np.zeros(15, dtype=np.uint64)
    sub = os.environ.get("SPHMI_RCCL_LIB")
    if sub:
        # the substitute library's own counters AFTER the handle is gone (ncclCommDestroy checks for unreceived messages and open groups):
        # dlopen of the same path returns the copy libsphmi bound
        eng.close()
        m = C.CDLL(sub)
        if hasattr(m, "mockrccl_stats"):
            m.mockrccl_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
            v = (C.c_uint64 * 15)()
            m.mockrccl_stats(v, 15)
            mock = np.array(list(v), dtype=np.uint64)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), prog=np.array(prog, dtype=np.float64), info=info, mock=mock, **d)


if __name__ == "__main__":
    main()
