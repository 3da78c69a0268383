#!/usr/bin/env python3
"""bench.py — particle-updates/s of the SPH hot path on the 3-D dam break (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W                        (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full time step of the hot path (cell-list upkeep + both neighbour passes with the
fused predictor / corrector + the Δt/Δx reductions) over every particle.  Inputs are generated on the
host and uploaded BEFORE the timed region; the timed region is exactly K steps bracketed by a barrier +
torch.cuda.synchronize() on both sides; the reported time is the max over ranks.

Workload: BASELINE config 3 — synthetic 3-D dam break at dp = 0.00425 (≈1.06 M particles), fp32
kernels, parameters of example/Dambreak3d.jl.  For N > 1 the lattice is refined so that every GPU keeps
≈1.06 M particles (weak scaling; N = 8 is BASELINE config 4, dp = 0.002125, ≈7.7 M particles); the slabs are driven by
the slab driver inside libsphmi.so (sphmi_create_rank: one slab per process, peers over RCCL).

Extra objects on the JSON line:
  roofline     — dominant kernel (k_neighbor_force): ALGORITHMIC bytes per launch ÷ its average launch
                 duration (HIP events on the engine's stream) against the 8 TB/s HBM peak.
                 Algorithmic bytes: (11·D+5)·4+2 = 154 B per particle-update = 77 B per
                 particle per launch (two launches per update).  The kernel is bound by vector-ALU issue
                 (≈1.1 k distance tests + ≈174 pair evaluations per particle per launch), so `frac` is
                 small; `valu` reports the binding resource from the committed counters of the shipped kernel.
  cpu_baseline — the CPU oracle (OpenMP restatement of the reference algorithm, fp64, "port") timed on
                 this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_UPDATE_3D_FP32 = (11 * 3 + 5) * 4 + 2      # 154 B, SURVEY.md §8d
# vector-ALU issue peak of the data sheet: 256 CUs × 4 SIMD-32, one wave64 instruction per 2 cycles each at 2.4 GHz (= the
# 157.3 TFLOP/s fp32 vector peak when every instruction is an FMA; MI355X_MICROARCH.md).  The SQ "busy" counter charges a
# quad-cycle per instruction instead (measured issue cost of most of this kernel's instructions: tools/ubench/valu_rates2.hip).
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
TRAFFIC_RECORD = "profiles/r02_hbm_traffic.json"
VALU_RECORD = "profiles/r02_valu_counters.json"


def measured_traffic(n_local):
    """Bytes per launch of the dominant kernel that leave the L2s, from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled — MI355X_MICROARCH.md's gfx950 correction, re-calibrated in the same passes on two kernels
    of known traffic — plus WRITE_SIZE), scaled from the profiled particle count.  None when the record is missing."""
    try:
        rec = json.load(open(os.path.join(ROOT, TRAFFIC_RECORD)))
        return rec["bytes_per_particle_per_launch_corrected"] * n_local
    except Exception:
        return None


def valu_from_counters(n_local, kern_ms):
    """The binding resource of the kernel, from COUNTERS (profiles/r02_valu_counters.json: SQ_INSTS_VALU and
    SQ_ACTIVE_INST_VALU of the shipped kernel, rocprofv3 --pmc): vector instructions per launch scaled from the profiled
    particle count ÷ the launch duration measured live, against the issue peak of the chip; `busy_frac_pmc` is the
    fraction of SIMD cycles with a vector instruction executing in the profiled run itself."""
    try:
        rec = json.load(open(os.path.join(ROOT, VALU_RECORD)))
        f = rec["final"]
        insts = 0.5 * (f["predictor"]["valu_insts"] + f["corrector"]["valu_insts"]) * n_local / rec["n_particles"]
        rate = insts / (kern_ms * 1e-3) if kern_ms > 0 else 0.0
        return {"wave_insts_per_launch": insts, "issue_rate": rate, "issue_peak": VALU_ISSUE_PEAK, "frac": rate / VALU_ISSUE_PEAK,
                "busy_frac_pmc": 0.5 * (f["predictor"]["valu_busy_frac"] + f["corrector"]["valu_busy_frac"]),
                "unit": "wave64 vector instructions/s", "source": VALU_RECORD}
    except Exception:
        return None


def measured_copy_bandwidth(device, seconds=0.08):
    """the HBM fraction is to be reported against the nominal 8 TB/s AND against what a copy kernel reaches
    on this box.  Read + write bytes per second of a 1 GiB device-to-device copy (torch's copy kernel on the current
    stream), measured live before the warm-up steps: a few warm-up copies, then repetitions for ≈`seconds`.
    (A side effect worth knowing: the device has left its idle clocks when the warm-up steps begin — on a cold MI355X the
    first ≈30 ms of neighbour-kernel launches run 583 → 490 µs, profiles/r02_cold_start_timeline.md.)"""
    import torch
    n = 1 << 28                                    # 1 GiB of float32 per buffer
    a = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    b.copy_(a)
    torch.cuda.synchronize(device)
    one = max(time.perf_counter() - t0, 1e-5)
    reps = max(4, int(seconds / one))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    gbs = 2 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return gbs


def precondition(device, ms, setup_fn, make_fn):
    """Untimed device pre-conditioning, BEFORE the contract's W warm-up steps and on a SCRATCH handle (a second engine with
    the generated 1.06 M-particle lattice; the measured handle executes exactly W + K steps).  Why: the MI355X drops to a
    low-clock state within 50 ms of idle and needs ≈25–30 ms of vector-ALU-bound load to come back — after import torch, the
    lattice set-up or a 0.05 s pause the neighbour kernel runs 0.55–0.58 ms per launch and reaches its steady 0.49 ms only
    ≈25 steps later (tools/two_engines.py, profiles/r02_cold_start_timeline.md; a memory-bound copy load does not lift it).
    A 20-step window right after 5 warm-up steps would measure the governor, not the engine.  `--precondition-ms 0` switches
    it off.  Returns the scratch engine (kept alive until the end: freeing it would put an idle gap before the warm-up)."""
    if ms <= 0:
        return None, 0
    dp1 = 0.00425
    scratch = make_fn(dp1, setup_fn(dp1), device_float_bytes=4, device=device)
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        scratch.advance(1e9, max_steps=16)
        n += 16
    return scratch, n


def cpu_baseline(dp=0.00425, steps=12):
    """Bounded CPU sample of THE BENCH WORKLOAD: the same 1.06 M-particle lattice (dp = 0.00425), `steps` steps after
    the step that holds the one-off sort (≈15 s).  The restatement keeps the reference's nthreads full-length
    accumulator copies (src/PreProcess.jl:204-205), so more threads is not always faster: the thread count is
    picked by a 1-step probe over {cores, cores/2, cores/4, 16, 8}."""
    from oracle.oracle import Oracle, make_oracle
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    cores = min(os.cpu_count() or 1, Oracle.max_threads())
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    o = make_oracle(p, s, threads=cores)
    o.advance(1e9, max_steps=1)                       # first step holds the one-off sort
    best, best_t = cores, float("inf")
    for t in sorted({cores, max(cores // 2, 1), max(cores // 4, 1), min(16, cores), min(8, cores)}, reverse=True):
        o.set_threads(t)
        t0 = time.perf_counter()
        o.advance(1e9, max_steps=1)
        el = time.perf_counter() - t0
        if el < best_t:
            best, best_t = t, el
    o.set_threads(best)
    t0 = time.perf_counter()
    pr = o.advance(1e9, max_steps=steps)
    dt = time.perf_counter() - t0
    return {"value": len(p) * pr.steps_done / dt, "unit": "particle-updates/s", "cores": best,
            "kind": "port",
            "sample": f"3-D dam break dp={dp} (N={len(p)}), {pr.steps_done} steps, fp64 OpenMP restatement "
                      f"of the reference algorithm (oracle/sph_oracle.c) on {best} of {cores} host threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dp", type=float, default=None, help="override lattice spacing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precondition-ms", type=float, default=60.0,
                    help="untimed device pre-conditioning on a scratch handle before the warm-up steps (0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (cold window, fp64, developed flow, parity)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="use the slab driver even for one rank (measures its host overhead)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libsphmi has no CPU path")
    # One rank per GPU is the contract.  With MORE ranks than GPUs (the one-GPU test box: `torchrun --nproc-per-node 2
    # bench.py --gpus 2` exercises this file's multi-rank path end to end) the ranks share devices, which RCCL refuses:
    # the slab driver then talks through its host shared-memory transport (SPHMI_TRANSPORT=shm) and the line says so.
    n_dev = torch.cuda.device_count()
    device = local_rank % n_dev
    if world > n_dev:
        os.environ.setdefault("SPHMI_TRANSPORT", "shm")
    torch.cuda.set_device(device)

    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    dp1 = 0.00425
    dp = args.dp or dp1 / (world ** (1.0 / 3.0))
    setup = setup_dam_break_3d(dp)

    from sphexample_amd.engine import dam_break_3d_count, make_engine, make_generated_dam_break_engine, rccl_unique_id
    info = None
    if world == 1 and not args.force_distributed:
        # the lattice is generated ON THE DEVICE (sphmi_generate_dam_break_3d, SURVEY §8 f4: identical to the host generator's
        # upload, tests/test_engine_gpu.py::test_device_side_case_generator) — nothing of it exists on the host
        n_total = sum(dam_break_3d_count(dp))
        eng = make_generated_dam_break_engine(dp, setup, device_float_bytes=4, device=device)
        barrier = lambda: None  # noqa: E731
        reduce_max = lambda x: x  # noqa: E731
    else:
        # One process per GPU (the launch contract).  torch.distributed is the RENDEZVOUS only (gloo: the RCCL unique id,
        # the barriers and the max over ranks of the wall time); halos, migration and the per-step MAX-allreduce run
        # inside libsphmi.so on RCCL (csrc/sphmi_multi.h, sphmi_create_rank).  Every rank generates the deterministic
        # lattice and keeps its slab.
        import torch.distributed as dist
        particles = dam_break_3d(dp)
        n_total = len(particles)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        def create():
            # rank 0 makes the 128-byte id (RCCL's, or any random bytes for the shared-memory transport), everybody gets it
            if rank == 0:
                try:
                    uid = [rccl_unique_id() if os.environ.get("SPHMI_TRANSPORT") != "shm" else os.urandom(128)]
                except Exception as exc:                      # librccl missing: say so and let the fallback below decide
                    uid = [os.urandom(128)]
                    print(f"[bench] rccl_unique_id failed: {exc}", file=sys.stderr)
            else:
                uid = [None]
            dist.broadcast_object_list(uid, src=0)
            try:
                return make_engine(particles, setup, device_float_bytes=4, device=device, rank=rank, world=world, unique_id=uid[0]), ""
            except Exception as exc:
                return None, str(exc)

        eng, why = create()
        ok = torch.tensor([1 if eng is not None else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()) and os.environ.get("SPHMI_TRANSPORT") != "shm":
            # RCCL could not be set up on some rank (every rank learns it here): the run still produces a line, over the host
            # shared-memory transport of the node — labelled in `parallelism`, not a valid multi-GPU measurement
            print(f"[bench] rank {rank}: RCCL set-up failed ({why or 'on another rank'}); falling back to SPHMI_TRANSPORT=shm", file=sys.stderr)
            del eng
            os.environ["SPHMI_TRANSPORT"] = "shm"
            eng, why = create()
            ok = torch.tensor([1 if eng is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            raise SystemExit(f"bench.py: rank {rank} could not create its slab engine: {why}")
        info = eng.multi_info()
        barrier = dist.barrier

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

    copy_gbs = measured_copy_bandwidth(torch.device("cuda", device))
    scratch, pre_steps = precondition(device, args.precondition_ms, setup_dam_break_3d, make_generated_dam_break_engine)
    eng.advance(1e9, max_steps=args.warmup)
    eng.force_kernel_stats(reset=True)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    prog = eng.advance(1e9, max_steps=args.steps)
    torch.cuda.synchronize(); barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    assert prog.steps_done == args.steps
    kern_ms, kern_launches = eng.force_kernel_stats()

    if rank == 0:
        value = n_total * args.steps / elapsed
        n_local = n_total / world
        alg_bytes_launch = BYTES_PER_UPDATE_3D_FP32 / 2.0 * n_local
        achieved = alg_bytes_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "particle-updates/sec (3D dam-break)", "value": value, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D dam-break dp={dp:.6g}, N={n_total} particles, "
                                   f"example/Dambreak3d.jl parameters, fp32 kernels",
                       "particles": n_total, "particles_per_gpu": n_local,
                       "parallelism": "single GPU" if world == 1 else
                       f"{'xyz'[info.axis]}-slab domain decomposition x{world} inside libsphmi.so, 1-cell halo over " +
                       ("RCCL (ncclSend/ncclRecv between slab neighbours + one 4-word ncclAllReduce per step), interior tiles overlap the exchange"
                        if info.transport == 1 else
                        f"the HOST SHARED-MEMORY transport ({world} ranks on {n_dev} GPU(s); messages staged through the host: not a valid multi-GPU measurement)"),
                       "rebuilds_in_window": int(prog.n_rebuilds), "sim_time": prog.total_time,
                       "preconditioning": (f"{pre_steps} untimed steps of a scratch handle (≈{args.precondition_ms:.0f} ms of the same kernels) before the "
                                           f"{args.warmup} warm-up steps: clock governor out of its idle state; the measured handle ran {args.warmup} + {args.steps} steps")
                       if pre_steps else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "peak_measured": copy_gbs, "frac_of_measured": achieved / copy_gbs if copy_gbs > 0 else None,
                         "peak_measured_source": "1 GiB device-to-device copy (read + write bytes) on this GPU, measured in this run before the warm-up steps",
                         "traffic": measured_traffic(n_local),
                         "traffic_source": TRAFFIC_RECORD + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bytes leaving the L2s, Infinity-Cache hits included)",
                         "kernel": "k_neighbor_force", "avg_launch_ms": kern_ms, "launches": kern_launches,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "valu": valu_from_counters(n_local, kern_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    del scratch
    if world > 1 or args.force_distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
