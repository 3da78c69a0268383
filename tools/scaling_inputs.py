#!/usr/bin/env python3
"""Inputs of the pre-registered scaling prediction, measured on ONE GPU.

For N = 1, 2, 4, 8 — bench.py's weak-scaling workloads: the 3-D dam break at dp = 0.00425 / N^(1/3), N = 8 is BASELINE config 4 —
the N-slab decomposition is built in ONE handle on GPU 0 (make_engine(devices=[0] * N): same planner, same cuts, same ghost layers and
tile lists as N GPUs would hold) and advanced; per slab it records
  * rows held (owned + ghost copies) and the halo records it sends per face with state A and with the half-step state H
    (sphmi_multi_halo_info; a record is 32 bytes in fp32) -> halo bytes per face and neighbour pass;
  * tiles of the interior launch and of the slab-edge launch -> share of edge tiles;
and per N the step time of the N slabs SERIALISED on the one chip next to N × the one-device step of the same per-GPU size:
what the slab machinery costs in device work (ghost rows scanned, split launches, pack / unpack, the local copies that stand in
for the xGMI transfers) before anything overlaps.

  python tools/scaling_inputs.py [steps] > gpurun_out/r06/scaling_inputs.json          (≈3 GPU-minutes)
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d  # noqa: E402
from sphexample_amd.engine import make_engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
DP1 = 0.00425


def step_ms(e, n):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); e.advance(1e9, max_steps=n); best = min(best, time.perf_counter() - t0)
    return best / n * 1e3


def halo_info(e, world):
    out = (C.c_int64 * (12 * world))()
    n = C.c_int32()
    e._lib.sphmi_multi_halo_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]
    e._check(e._lib.sphmi_multi_halo_info(e._h, out, 12 * world, C.byref(n)))
    keys = ("slab", "rows", "send_l_A", "send_r_A", "send_l_H", "send_r_H", "tiles_interior", "tiles_edge", "run_interior", "run_edge", "pass1_alone_ns", "pass2_alone_ns")
    return [dict(zip(keys, out[12 * k:12 * k + 12])) for k in range(n.value // 12)]


res = {"steps": steps, "record_bytes": 32, "runs": []}
one = make_engine(dam_break_3d(DP1), setup_dam_break_3d(DP1), device_float_bytes=4)
one.advance(1e9, max_steps=40)
t1 = step_ms(one, steps)
res["one_device_ms_per_step_1.06M"] = t1
del one
for world in (2, 4, 8):
    dp = DP1 / world ** (1.0 / 3.0)
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    e = make_engine(p, s, device_float_bytes=4, devices=[0] * world)
    e.advance(1e9, max_steps=20)
    ms = step_ms(e, steps)
    info = e.multi_info()
    slabs = halo_info(e, world)
    owned = e.owned_count()
    run = {"slabs": world, "dp": dp, "N": len(p), "axis": int(info.axis), "halo_width": int(info.halo_width), "cuts": [int(c) for c in info.cuts[:world - 1]],
           "ms_per_step_all_slabs_on_one_gpu": ms, "ms_per_step_per_slab": ms / world, "overhead_vs_one_device_step": ms / world / (t1 * (len(p) / world) / 1057738.0) - 1.0,
           "per_slab": []}
    for q in slabs:
        faces = (q["send_l_A"] > 0) + (q["send_r_A"] > 0)
        run["per_slab"].append(dict(q, ghost_rows=None, halo_bytes_per_face_pass1=32 * max(q["send_l_A"], q["send_r_A"]),
                                    halo_bytes_per_face_pass2=32 * max(q["send_l_H"], q["send_r_H"]), faces=faces,
                                    edge_tile_share=q["tiles_edge"] / max(q["tiles_edge"] + q["tiles_interior"], 1)))
    rows = sum(q["rows"] for q in slabs)
    run["ghost_rows_total"] = rows - owned
    run["ghost_share"] = (rows - owned) / owned
    res["runs"].append(run)
    print(f"[scaling] {world} slabs, N = {len(p)}: {ms:.3f} ms per step on one GPU = {ms / world:.3f} per slab (one device, 1.06 M: {t1:.3f}); "
          f"halo per face {max(x['halo_bytes_per_face_pass1'] for x in run['per_slab']) / 1e6:.2f} MB, edge tiles "
          f"{max(x['edge_tile_share'] for x in run['per_slab']) * 100:.1f} %, ghost rows {run['ghost_share'] * 100:.1f} %", file=sys.stderr, flush=True)
    del e
print(json.dumps(res, indent=1))
