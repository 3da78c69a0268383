"""One process of a rank-mode run (sphmi_create_rank): spawned `world` times by tests/test_rank_mode.py.

  python rank_worker.py selftest <unique id, hex> <rank> <world> <bytes per message>
  python rank_worker.py run <unique id, hex> <rank> <world> <case> <steps> <float bytes> <out dir> <advance calls> <slab axis | -1>

`run` creates slab `rank` of the case on GPU 0 with SPHMI_TRANSPORT=shm (the environment of the spawner) — or, with
$SPHMI_TEST_DEVICE_PER_RANK=1 (tests/test_multi_device_gpu.py on a box with several GPUs), on GPU `rank` over RCCL —
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
    eng = make_engine(p, s, device_float_bytes=fb, device=device, rank=rank, world=world, unique_id=uid,
                      slab_axis=None if axis < 0 else axis)
    prog = []
    for _ in range(calls):
        pr = eng.advance(1e9, max_steps=steps // calls)
        prog.append([pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter, pr.total_time, pr.last_dt])
    info = eng.multi_info()
    d = eng.download(("Position", "Density", "ID", "Velocity"))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), prog=np.array(prog, dtype=np.float64),
             info=np.array([info.world, info.n_local, info.transport, info.axis, info.halo_width, info.n_recuts, info.reserved]), **d)


if __name__ == "__main__":
    main()
