#!/usr/bin/env python3
"""bench.py — particle-updates/s of the SPH hot path on the 3-D dam break (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W                        (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W      (one rank per GPU)
  python bench.py --gpus N --steps K --warmup W                        (N > 1 WITHOUT a launcher: bench.py starts the N ranks
                                                                        itself — the same torch.distributed.run command line —
                                                                        and, if that cannot be done, falls into the one-process
                                                                        handle below; `config.launch` says which)
  python bench.py --gpus N --single-process --steps K --warmup W       (ONE process, one handle over N GPUs: sphmi_create
                                                                        with a device list — what the reference's single
                                                                        Julia process would run)

One "step" = one full time step of the hot path (cell-list upkeep + both neighbour passes with the fused predictor /
corrector + the Δt/Δx reductions) over every particle.  Inputs are resident in HBM BEFORE the timed region; the timed
region is exactly K steps bracketed by a barrier + torch.cuda.synchronize() on both sides; the reported time is the max
over ranks.

Workload: BASELINE config 3 — synthetic 3-D dam break at dp = 0.00425 (≈1.06 M particles), fp32 kernels, parameters of
example/Dambreak3d.jl.  For N > 1 the lattice is refined so that every GPU keeps ≈1.06 M particles (weak scaling; N = 8 is
BASELINE config 4, dp = 0.002125, ≈7.7 M particles); the slabs are driven by the slab driver inside libsphmi.so.
If RCCL cannot be set up in rank mode, EVERY rank learns it (sphexample_amd/rendezvous.py) and the run falls back, in this
order: rank 0 alone drives all N GPUs through one multi-device handle (RCCL ncclCommInitAll, then stream-ordered peer
copies) — still a multi-GPU measurement, labelled in `config.parallelism` — and only with more ranks than GPUs (the one-GPU
test box) the host shared-memory transport, which is labelled as not a valid multi-GPU measurement.

Objects on the JSON line beyond the contract's keys:
  roofline     — dominant kernel (k_neighbor_force): ALGORITHMIC bytes per launch ÷ its average launch duration (HIP events
                 on the engine's stream) against the 8 TB/s HBM peak: `achieved`, `peak`, `frac` are that HBM figure.
                 (11·D+5)·4+2 = 154 B per particle-update = 77 B per particle per launch.  `bound` names the
                 resource that actually binds the kernel — since the middle of round 5 the rate at which the texture path takes
                 per-lane gathers (`valu.gather`: wave-level gathers per launch and CU-cycles per gather next to the
                 micro-benchmark's figures), with vector-ALU issue ≈10 % behind (`valu`) — priced from the committed
                 counters of the shipped kernel.  Counters are only quoted when the ISA of the loaded library's two bench
                 kernels hashes to what the counter record was taken on (tools/isa_report.py): otherwise `traffic` and
                 `valu` are null and `counters_refused` says why.
  cpu_baseline — the CPU oracle (OpenMP restatement of the reference algorithm, fp64, "port") timed on this box's host cores
                 on a bounded sample of the same workload.
  value_cold   — the same W + K window on a handle that starts on an idle device (no pre-conditioning); `value` is measured
                 after `config.preconditioning` (the clock governor: profiles/r02_cold_start_timeline.md).
  value_excl_rebuild — the window with the cell-list rebuilds' device time taken out (every sphmi_advance opens with one:
                 the reference re-arms Δx at src/SPHCellList.jl:739).
  fp64, developed_window, parity — secondary measurements (the reference's own arithmetic; 200 steps at t ≈ 0.4 s; a short
                 fp32-vs-oracle check).  `--no-extras` skips them and value_cold.
  value_end_to_end / end_to_end — the reference's RunSimulation call pattern at the headline size: one sphmi_advance per 0.01 s of
                 simulated time, every call followed by the asynchronous download of all carried fields, t = 0 → 0.05 s on a cold
                 handle; updates/s over the whole run (host work between the calls included).
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
SHADER_CLOCK_HZ, N_CU = 2.4e9, 256
VALU_ISSUE_PEAK = N_CU * 4 * SHADER_CLOCK_HZ / 2.0
COUNTER_RECORD = "profiles/r06_counters.json"
DP1 = 0.00425
BENCH_KERNELS = {"predictor": "k_neighbor_force<float, 3, 1, 33, 2, 2>", "corrector": "k_neighbor_force<float, 3, 2, 33, 2, 2>"}


def loaded_kernel_identity():
    """ISA fingerprint, registers and LDS of the two neighbour kernels this benchmark launches, read from the code object
    of the library that is LOADED (sphexample_amd.build.LIB or $SPHMI_LIB) — tools/isa_report.py: llvm-objdump of the gfx950
    code object, comments stripped, sha256.  None when the LLVM tools are missing."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_report
        from sphexample_amd import build
        rep = isa_report.report(os.environ.get("SPHMI_LIB") or build.LIB, list(BENCH_KERNELS.values()))
        out = {}
        for role, name in BENCH_KERNELS.items():
            r = next(v for k, v in rep.items() if name in k)
            out[role] = {k: r[k] for k in ("symbol", "isa_sha16", "vgprs", "lds_bytes", "scratch_bytes", "instructions")}
            if "pair_loop" in r:
                out[role]["pair_loop_vector_alu"] = r["pair_loop"]["vector_alu_incl_trans"]
        return out
    except Exception as exc:                               # noqa: BLE001 — the line says why the counters are not quoted
        return {"error": f"{type(exc).__name__}: {exc}"}


def counters_for(identity, n_local, kern_ms):
    """(traffic bytes per launch, valu object, refusal text).  The record holds PMC counters of a profiled run (rocprofv3
    --pmc passes, tools/pmc_passes.sh → tools/pmc_derive.py) TOGETHER with the identity of the kernels they were taken on;
    they are quoted only for exactly those kernels."""
    try:
        rec = json.load(open(os.path.join(ROOT, COUNTER_RECORD)))
    except Exception as exc:                               # noqa: BLE001
        return None, None, f"{COUNTER_RECORD}: {exc}"
    if not identity or "error" in identity:
        return None, None, f"kernel identity of the loaded library unavailable ({(identity or {}).get('error', 'no report')})"
    for role in BENCH_KERNELS:
        want, got = rec.get("kernels", {}).get(role, {}), identity.get(role, {})
        for k in ("symbol", "isa_sha16", "vgprs", "lds_bytes"):
            if want.get(k) != got.get(k):
                return None, None, (f"{COUNTER_RECORD} was taken on another kernel: {role} {k} = {want.get(k)!r} in the record, "
                                    f"{got.get(k)!r} in the loaded library — re-run tools/pmc_passes.sh + tools/pmc_derive.py")
    c, n_prof = rec["counters"], rec["n_particles"]
    insts = 0.5 * (c["predictor"]["valu_insts"] + c["corrector"]["valu_insts"]) * n_local / n_prof
    rate = insts / (kern_ms * 1e-3) if kern_ms > 0 else 0.0
    valu = {"wave_insts_per_launch": insts, "issue_rate": rate, "issue_peak": VALU_ISSUE_PEAK, "frac": rate / VALU_ISSUE_PEAK,
            "busy_frac_pmc": 0.5 * (c["predictor"]["valu_busy_frac"] + c["corrector"]["valu_busy_frac"]),
            "waves_per_simd_pmc": 0.5 * (c["predictor"]["waves_per_simd_mean"] + c["corrector"]["waves_per_simd_mean"]),
            "wave_time_on_waitcnt_pmc": 0.5 * (c["predictor"]["wave_time_parked_on_waitcnt"] + c["corrector"]["wave_time_parked_on_waitcnt"]),
            "unit": "wave64 vector instructions/s", "source": COUNTER_RECORD}
    # The unit the launch runs out of since the middle of round 5 (profiles/HISTORY.md §4.9): the texture path charges a wave-level gather by its lanes and
    # segments, not by its bytes — tools/ubench/gather4.hip on this chip, CU-cycles per b128 instruction with the data in L1: 34 for 64 lanes
    # with scattered records inside 4 KB, 24.5 inside 256 B, 26 for 32 lanes, 17 coalesced.  The kernels' gathers have ≈55 of 64 lanes switched on
    # and, with six-entry queues, lanes that stay close together: ≈30 cycles.
    gathers = 0.5 * (c["predictor"].get("vmem_rd_insts", 0.0) + c["corrector"].get("vmem_rd_insts", 0.0)) * n_local / n_prof
    cyc = kern_ms * 1e-3 * SHADER_CLOCK_HZ * N_CU / gathers if gathers > 0 and kern_ms > 0 else 0.0
    if gathers > 0:
        valu["gather"] = {"wave_gathers_per_launch": gathers, "cu_cycles_per_gather": cyc,
                          "ubench_cu_cycles_per_gather": {"64 lanes scattered in 4 KB": 34.1, "64 lanes scattered in 256 B": 24.5, "32 lanes scattered": 26.5, "coalesced": 17.4},
                          "frac_of_ubench_rate_64_lanes": 34.1 / cyc if cyc > 0 else None,
                          "ta_busy_frac_pmc": 0.5 * (c["predictor"].get("ta_busy_frac", 0.0) + c["corrector"].get("ta_busy_frac", 0.0)),
                          "l1_hit_frac_pmc": 0.5 * (c["predictor"].get("l1_hit_frac", 0.0) + c["corrector"].get("l1_hit_frac", 0.0)),
                          "source": COUNTER_RECORD + ", profiles/r05_raw/gather4_l1.txt"}
    traffic = rec["traffic"]["bytes_per_particle_per_launch_corrected"] * n_local if "traffic" in rec else None
    return traffic, valu, None


def measured_copy_bandwidth(device, seconds=0.08):
    """the HBM fraction is to be reported against the nominal 8 TB/s AND against what a copy kernel reaches
    on this box.  Read + write bytes per second of a 1 GiB device-to-device copy (torch's copy kernel on the current
    stream), measured live before the warm-up steps."""
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


def timed_window(eng, warmup, steps, barrier=lambda: None, reduce_max=lambda x: x):
    """The contract's window on one handle: W untimed steps, then exactly K steps between barrier + synchronize pairs."""
    import torch
    eng.advance(1e9, max_steps=warmup)
    eng.force_kernel_stats(reset=True)
    t_before = dict(eng.timers())
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    prog = eng.advance(1e9, max_steps=steps)
    torch.cuda.synchronize(); barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    assert prog.steps_done == steps
    kern_ms, kern_launches = eng.force_kernel_stats()
    t_after = eng.timers()
    label = "02a Actual Calculate IndexCounter"                      # the rebuild phase under the reference's TimerOutputs label
    rebuild_s = t_after[label][0] - t_before[label][0]
    rebuilds = t_after[label][1] - t_before[label][1]
    return elapsed, prog, kern_ms, kern_launches, rebuild_s, rebuilds


def precondition(scratch, ms):
    """Untimed device pre-conditioning on the SCRATCH handle (the measured handle executes exactly W + K steps).  The MI355X
    drops to a low-clock state within 50 ms of idle and needs ≈25–30 ms of vector-ALU-bound load to come back
    (profiles/r02_cold_start_timeline.md); a 20-step window right after 5 warm-up steps measures the governor — that
    figure is reported too, as value_cold."""
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        scratch.advance(1e9, max_steps=16)
        n += 16
    return n


def cpu_baseline(dp=DP1, steps=12):
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


def extra_fp64(device, warmup, steps):
    """The reference computes in Float64 (every stock example): the same window with fp64 kernels (device_float_bytes = 8)."""
    from sphexample_amd.cases import setup_dam_break_3d
    from sphexample_amd.engine import dam_break_3d_count, make_generated_dam_break_engine
    eng = make_generated_dam_break_engine(DP1, setup_dam_break_3d(DP1), device_float_bytes=8, device=device)
    el, prog, kms, _, _, _ = timed_window(eng, warmup, steps)
    n = sum(dam_break_3d_count(DP1))
    eng.close()
    return {"value": n * steps / el, "unit": "particle-updates/s", "ms_per_step": el / steps * 1e3, "dtype": "f64",
            "kernel_avg_launch_ms": kms, "steps": steps, "warmup": warmup}


def extra_developed(device, t_target=0.4, window=200):
    """"also report a window at t ≈ 0.4 s" — the front has hit the pillar and the far wall, spray tiles span
    hundreds of cells, the cell list is rebuilt every ≈30 steps."""
    import torch
    from sphexample_amd.cases import setup_dam_break_3d
    from sphexample_amd.engine import dam_break_3d_count, make_generated_dam_break_engine
    eng = make_generated_dam_break_engine(DP1, setup_dam_break_3d(DP1), device_float_bytes=4, device=device)
    n = sum(dam_break_3d_count(DP1))
    t0 = time.perf_counter(); pr = eng.advance(t_target); torch.cuda.synchronize(); t1 = time.perf_counter()
    r0, it0 = pr.n_rebuilds, pr.iteration
    eng.force_kernel_stats(reset=True)
    t2 = time.perf_counter(); pw = eng.advance(1e9, max_steps=window); torch.cuda.synchronize(); t3 = time.perf_counter()
    kms, _ = eng.force_kernel_stats()
    eng.close()
    return {"value": n * window / (t3 - t2), "unit": "particle-updates/s", "ms_per_step": (t3 - t2) / window * 1e3,
            "steps": window, "sim_time": pw.total_time, "rebuilds_in_window": int(pw.n_rebuilds - r0), "kernel_avg_launch_ms": kms,
            "run_up": {"steps": int(it0), "rebuilds": int(r0), "seconds": t1 - t0, "value": n * it0 / (t1 - t0)}}


def extra_end_to_end(device, t_end=0.05, interval=0.01):
    """What a `RunSimulation` of the reference sees at the headline size (src/SPHCellList.jl:881-929): ONE sphmi_advance per output interval
    (`OutputTimes` = 0.01 s of simulated time, example/Dambreak3d.jl) — each opening with a cell-list rebuild, each followed by the download
    of every SimParticles field the engine carries into the caller's Float64 arrays (asynchronously: the copies of interval k travel while
    interval k + 1 is computed, sphmi_download_begin / _end) and by the sort's permutation for the passive columns — from t = 0 to `t_end`,
    on a handle that starts cold.  updates/s over the WHOLE run, host work between the calls included: where `value` (warm clock, one call)
    and `value_cold` bracket what a user gets.  The full 0 → 0.4 s run: tools/examples_end_to_end.py dam_break_3d_c3, profiles/r06_examples_end_to_end.md."""
    import types

    import numpy as np
    import torch
    from sphexample_amd.cases import setup_dam_break_3d
    from sphexample_amd.engine import dam_break_3d_count, make_generated_dam_break_engine
    eng = make_generated_dam_break_engine(DP1, setup_dam_break_3d(DP1), device_float_bytes=4, device=device)
    n = sum(dam_break_3d_count(DP1))
    P = types.SimpleNamespace(Position=np.zeros((n, 3)), Velocity=np.zeros((n, 3)), Acceleration=np.zeros((n, 3)), Density=np.zeros(n), Pressure=np.zeros(n),
                              ID=np.zeros(n, dtype=np.int64), Type=np.zeros(n, dtype=np.uint8), GroupMarker=np.zeros(n, dtype=np.uint64),
                              GhostPoints=np.zeros((n, 3)), Cells=np.zeros((n, 3), dtype=np.int64))
    eng.pin(P)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k, pending, dl = 1, False, 0.0
    while True:
        prog = eng.advance(interval * k)                    # next_output_time(SimMetaData) = OutputTimes × counter, :687-698
        td = time.perf_counter()
        if pending:
            eng.download_end()
        eng.download_into_begin(P); pending = True
        eng.download_permutation()
        dl += time.perf_counter() - td
        k += 1
        if prog.total_time > t_end:
            break
    eng.download_end()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"value": n * prog.iteration / el, "unit": "particle-updates/s", "steps": int(prog.iteration), "intervals": k - 1, "sim_time": prog.total_time,
           "seconds": el, "host_seconds_in_output_calls": dl, "rebuilds": int(prog.n_rebuilds),
           "pattern": f"one sphmi_advance per {interval} s of simulated time + asynchronous download of all carried fields into page-locked Float64 arrays + sphmi_download_permutation"}
    eng.unpin(); eng.close()
    return out


def extra_parity(device, dp=0.0085, steps=10):
    """A short parity check INSIDE the bench run: fp32 kernels vs the fp64 oracle on the reference example's own resolution
    (example/Dambreak3d.jl: dp = 0.0085, ≈150 k particles), `steps` steps from a perturbed state — the tolerance of the
    north star is 1e-5 on density and position (tests/test_config_scale_gpu.py holds the 1 M / 7.7 M cases)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import perturbed
    from oracle.oracle import Oracle, make_oracle
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    p, s = perturbed(dam_break_3d(dp), seed=5), setup_dam_break_3d(dp)
    eng = make_engine(p, s, device_float_bytes=4, device=device)
    orc = make_oracle(p, s, threads=max(1, min(16, os.cpu_count() or 1, Oracle.max_threads())))
    pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
    e, o = eng.download(("ID", "Density", "Position")), orc.download(("ID", "Density", "Position"))
    ie, io = np.argsort(e["ID"], kind="stable"), np.argsort(o["ID"], kind="stable")
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
    out = {"dp": dp, "particles": len(p), "steps": steps, "rho": rel(e["Density"][ie], o["Density"][io]),
           "x": rel(e["Position"][ie], o["Position"][io]),
           "x_over_dp": float(np.abs(e["Position"][ie] - o["Position"][io]).max() / dp),
           "dt": abs(pe.last_dt - po.last_dt) / po.last_dt, "same_rebuilds": pe.n_rebuilds == po.n_rebuilds,
           "tolerance": 1e-5, "against": "oracle/sph_oracle.c (fp64 restatement of the reference algorithm)"}
    eng.close(); orc.close()
    return out


def parallelism_text(world, info, n_dev, how):
    if world == 1:
        return "single GPU"
    base = f"{'xyz'[info.axis]}-slab domain decomposition x{world} inside libsphmi.so, 1-cell halo over "
    tr = {1: "RCCL (ncclSend/ncclRecv between slab neighbours on one communicator, one 4-word ncclAllReduce per step on a second), "
             "interior tiles overlap the exchange",
          0: "stream-ordered device-to-device copies between the slabs of one process (peer access over xGMI when the slabs sit on different GPUs)",
          2: f"the HOST SHARED-MEMORY transport ({world} ranks on {n_dev} GPU(s); messages staged through the host: not a valid multi-GPU measurement)"}[info.transport]
    note = ""
    if info.transport == 1 and os.environ.get("SPHMI_RCCL_LIB"):
        # a line produced over a stand-in for librccl (tests/mock_rccl: several ranks on one GPU) must say so
        note = (f"; librccl SUBSTITUTED by $SPHMI_RCCL_LIB={os.environ['SPHMI_RCCL_LIB']} ({world} ranks on {n_dev} GPU(s)): a test of the launch, "
                "not a valid multi-GPU measurement")
    return base + tr + f"; {how}" + note


def self_spawn(world, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks here, with the command
    line the contract names (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py …), and pass their output through.  Returns the launcher's exit code."""
    import subprocess
    from sphexample_amd.rendezvous import free_port
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, SPHMI_BENCH_LAUNCH=f"self-spawned: bench.py was started without a launcher (WORLD_SIZE unset) and ran "
                                              f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node={world} --master-addr 127.0.0.1 "
                                              f"--master-port {port} bench.py …` itself — one rank per GPU, as the contract's launch does")
    print("[bench] no launcher (WORLD_SIZE unset): " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dp", type=float, default=None, help="override lattice spacing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (cold window, fp64, developed flow, parity)")
    ap.add_argument("--precondition-ms", type=float, default=60.0,
                    help="untimed device pre-conditioning on a scratch handle before the warm-up steps (0 = off: value = value_cold)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process: one multi-device handle (sphmi_create with a device list) instead of one rank per GPU")
    ap.add_argument("--force-distributed", action="store_true",
                    help="use the slab driver even for one rank (measures its host overhead)")
    args = ap.parse_args()

    launch = os.environ.get("SPHMI_BENCH_LAUNCH", "")
    spawn_failure = None
    if args.gpus > 1 and not args.single_process and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # Started like the N = 1 line (`python bench.py --gpus N …`): the ranks are started here; a launcher that cannot be
        # started (or ranks that fail) leaves the one-process handle over the N GPUs — a line comes out either way.
        rc = self_spawn(args.gpus, sys.argv[1:])
        if rc == 0:
            return
        print(f"[bench] the self-spawned ranks ended with exit code {rc}: falling back to ONE process with one handle over "
              f"{args.gpus} devices (--single-process)", file=sys.stderr, flush=True)
        args.single_process = True
        launch = (f"one process, one multi-device handle: bench.py was started without a launcher, its self-spawned "
                  f"torch.distributed.run ranks failed (exit code {rc})")
        spawn_failure = rc
    import torch
    rank = int(os.environ.get("RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libsphmi has no CPU path")
    n_dev = torch.cuda.device_count()
    world = args.gpus
    if args.single_process:
        if env_world != 1:
            raise SystemExit("--single-process runs in one process: do not launch it with torch.distributed.run")
    elif env_world != world:
        raise SystemExit(f"--gpus {world} but WORLD_SIZE={env_world}: launch with torch.distributed.run (or pass --single-process)")
    # One rank per GPU is the contract.  With MORE ranks than GPUs (the one-GPU test box: `torchrun --nproc-per-node 2
    # bench.py --gpus 2` exercises this file's multi-rank path end to end) the ranks share devices, which RCCL refuses:
    # the slab driver then talks through its host shared-memory transport (SPHMI_TRANSPORT=shm) and the line says so.
    device = local_rank % n_dev
    if env_world > n_dev:
        os.environ.setdefault("SPHMI_TRANSPORT", "shm")
    torch.cuda.set_device(device)

    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import dam_break_3d_count, make_engine, make_generated_dam_break_engine
    dp = args.dp or DP1 / (world ** (1.0 / 3.0))
    setup = setup_dam_break_3d(dp)
    identity = loaded_kernel_identity() if rank == 0 else None

    # ---- the scratch handle: the cold window (value_cold), then the pre-conditioning load -------------------------------
    cold = None
    scratch = None
    pre_steps = 0
    plain = world == 1 and not args.force_distributed
    if plain and (args.precondition_ms > 0 or not args.no_extras):
        scratch = make_generated_dam_break_engine(DP1, setup_dam_break_3d(DP1), device_float_bytes=4, device=device)
        if not args.no_extras and not args.dp:
            el, _, kms, _, _, _ = timed_window(scratch, args.warmup, args.steps)
            cold = {"value": sum(dam_break_3d_count(DP1)) * args.steps / el, "ms_per_step": el / args.steps * 1e3, "kernel_avg_launch_ms": kms}

    info, rdv, how = None, None, ""
    barrier = lambda: None  # noqa: E731
    reduce_max = lambda x: x  # noqa: E731
    active = True                    # does this process hold (part of) the measured simulation?
    if plain:
        # the lattice is generated ON THE DEVICE (sphmi_generate_dam_break_3d, SURVEY §8 f4: identical to the host generator's
        # upload, tests/test_engine_gpu.py::test_device_side_case_generator) — nothing of it exists on the host
        n_total = sum(dam_break_3d_count(dp))
        eng = make_generated_dam_break_engine(dp, setup, device_float_bytes=4, device=device)
    elif args.single_process:
        particles = dam_break_3d(dp)
        n_total = len(particles)
        devs = [k % n_dev for k in range(world)]
        eng = make_engine(particles, setup, device_float_bytes=4, devices=devs)
        info = eng.multi_info()
        how = f"one process, one handle over devices {devs} (--single-process)"
    else:
        # One process per GPU (the launch contract).  torch.distributed is the RENDEZVOUS only (gloo: the RCCL unique id,
        # the barriers, the max over ranks of the wall time, and "did every rank get its engine"); halos, migration and the
        # per-step MAX-allreduce run inside libsphmi.so on RCCL (csrc/sphmi_multi.h, sphmi_create_rank).  Every rank generates
        # the deterministic lattice and keeps its slab.
        from sphexample_amd.rendezvous import Rendezvous, create_rank_engine
        particles = dam_break_3d(dp)
        n_total = len(particles)
        rdv = Rendezvous(rank, env_world)
        barrier, reduce_max = rdv.barrier, rdv.max
        make = lambda uid: make_engine(particles, setup, device_float_bytes=4, device=device, rank=rank, world=world, unique_id=uid)  # noqa: E731
        # a communicator set-up that cannot reach its peers does not fail, it waits: give up loudly instead of sitting in the
        # driver's time limit ($SPHMI_BENCH_SETUP_TIMEOUT seconds, default 300)
        import threading

        def give_up():
            print(f"[bench] rank {rank}: the slab engines were not set up within the time limit (RCCL communicator set-up hanging?) — "
                  f"try `bench.py --gpus {world} --single-process` or SPHMI_TRANSPORT=local", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(float(os.environ.get("SPHMI_BENCH_SETUP_TIMEOUT", "300")), give_up)
        watchdog.daemon = True
        watchdog.start()
        eng, errs = create_rank_engine(rdv, make)
        watchdog.cancel()
        how = "one process per GPU (sphmi_create_rank)"
        if eng is None and os.environ.get("SPHMI_TRANSPORT") != "shm":
            if rank == 0:
                print("[bench] rank-mode RCCL set-up failed: " + " | ".join(errs), file=sys.stderr)
            if world <= n_dev:
                # every rank knows; rank 0 drives all GPUs through ONE multi-device handle, the others wait at the barriers
                how = ("FALLBACK: rank-mode RCCL set-up failed (" + "; ".join(errs)[:300] + "); rank 0 alone drives all GPUs through one "
                       "multi-device handle, the other ranks idle")
                active = rank == 0
                ok = True
                if active:
                    eng = None
                    for tr in (None, "local"):                            # RCCL inside one process, then plain peer copies
                        try:
                            if tr:
                                os.environ["SPHMI_TRANSPORT"] = tr
                            eng = make_engine(particles, setup, device_float_bytes=4, devices=list(range(world)))
                            break
                        except Exception as exc:                          # noqa: BLE001
                            print(f"[bench] one-process handle over {world} GPUs ({tr or 'rccl'}) failed: {exc}", file=sys.stderr)
                    ok = eng is not None
                if not rdv.all_ok(ok):
                    raise SystemExit("bench.py: no multi-GPU path could be set up (see stderr)")
            else:
                os.environ["SPHMI_TRANSPORT"] = "shm"
                eng, errs = create_rank_engine(rdv, make)
                how = "one process per GPU (sphmi_create_rank); RCCL set-up failed, shared-memory transport"
        if active and eng is None:
            raise SystemExit(f"bench.py: rank {rank} could not create its slab engine: {' | '.join(errs)}")
        if active:
            info = eng.multi_info()

    copy_gbs = measured_copy_bandwidth(torch.device("cuda", device)) if rank == 0 else 0.0
    if scratch is not None and args.precondition_ms > 0:
        pre_steps = precondition(scratch, args.precondition_ms)
    run_watchdog = None
    if world > 1:
        # the same for the steps themselves: a halo exchange or an allreduce whose peer never arrives waits for ever — say so and
        # leave instead of sitting in the driver's time limit ($SPHMI_BENCH_RUN_TIMEOUT seconds, default 900)
        import threading

        def give_up_run():
            print(f"[bench] rank {rank}: warm-up + {args.steps} steps did not finish within the time limit (a collective waiting for a peer?) — "
                  f"try `bench.py --gpus {world} --single-process` or SPHMI_TRANSPORT=shm", file=sys.stderr, flush=True)
            os._exit(4)
        run_watchdog = threading.Timer(float(os.environ.get("SPHMI_BENCH_RUN_TIMEOUT", "900")), give_up_run)
        run_watchdog.daemon = True
        run_watchdog.start()
    if active:
        elapsed, prog, kern_ms, kern_launches, rebuild_s, rebuilds = timed_window(eng, args.warmup, args.steps, barrier, reduce_max)
    else:
        barrier(); torch.cuda.synchronize(); torch.cuda.synchronize(); barrier()
        reduce_max(0.0)
    if run_watchdog is not None:
        run_watchdog.cancel()

    if rank == 0:
        value = n_total * args.steps / elapsed
        n_local = n_total / world
        alg_bytes_launch = BYTES_PER_UPDATE_3D_FP32 / 2.0 * n_local
        achieved = alg_bytes_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic, valu, refused = counters_for(identity, n_local, kern_ms)
        out = {
            "metric": "particle-updates/sec (3D dam-break)", "value": value,
            # the same W + K window on a handle that starts on an idle device (no pre-conditioning) — next to `value`, so that nobody reads one without the other
            "value_cold": cold["value"] if cold else (value if not pre_steps else None),
            "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D dam-break dp={dp:.6g}, N={n_total} particles, "
                                   f"example/Dambreak3d.jl parameters, fp32 kernels",
                       "particles": n_total, "particles_per_gpu": n_local,
                       "parallelism": parallelism_text(world, info, n_dev, how),
                       "launch": launch or ("torch.distributed.run (RANK / WORLD_SIZE from the launcher)" if env_world > 1 else "one process"),
                       "rebuilds_in_window": int(rebuilds), "sim_time": prog.total_time,
                       "preconditioning": (f"{pre_steps} untimed steps of a scratch handle (≈{args.precondition_ms:.0f} ms of the same kernels) before the "
                                           f"{args.warmup} warm-up steps: clock governor out of its idle state; the measured handle ran {args.warmup} + {args.steps} "
                                           f"steps; value_cold is the same window without it") if pre_steps else "none"},
            "roofline": {"bound": "gather_issue", "hbm_note": "achieved / peak / frac are the HBM figure BASELINE.json's metric asks for (algorithmic bytes ÷ kernel time "
                                                              "÷ 8 TB/s); the kernel is bound by the rate at which the texture path takes per-lane gathers — `valu.gather` — "
                                                              "with vector-ALU issue ≈10 % behind (`valu`; until the middle of round 5 it was the other way round)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "peak_measured": copy_gbs, "frac_of_measured": achieved / copy_gbs if copy_gbs > 0 else None,
                         "peak_measured_source": "1 GiB device-to-device copy (read + write bytes) on this GPU, measured in this run before the warm-up steps",
                         "traffic": traffic,
                         "traffic_source": COUNTER_RECORD + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bytes leaving the L2s, Infinity-Cache hits included)",
                         "kernel": "k_neighbor_force", "avg_launch_ms": kern_ms,
                         # two launches per step (predictor, corrector); the average is taken over HIP-event pairs on every 8th
                         # step (an event pair costs a stream bubble), each weighted 8: `launch_time_samples` is that weighted count
                         # `launches_expected`: predictor + corrector of every step of the window (edge-list launches of slab handles and steps the
                         # device-side control cancelled are not counted); `launch_time_samples`: the weighted count avg_launch_ms was averaged over
                         "launches_expected": 2 * args.steps, "launch_time_samples": kern_launches,
                         "algorithmic_bytes_per_launch": alg_bytes_launch, "valu": valu, "kernel_identity": identity},
            "value_excl_rebuild": n_total * args.steps / max(elapsed - rebuild_s, 1e-9),
            "rebuild_ms_in_window": rebuild_s * 1e3,
        }
        if refused:
            out["roofline"]["counters_refused"] = refused
        if cold:
            out["cold_window"] = cold
        if spawn_failure is not None:
            out["fallback_from_failed_ranks_exit_code"] = spawn_failure        # (the line below was measured by ONE process: its ranks could not be started)
        if plain and not args.no_extras and not args.dp:
            for name, fn in (("fp64", lambda: extra_fp64(device, args.warmup, args.steps)),
                             ("developed_window", lambda: extra_developed(device)), ("end_to_end", lambda: extra_end_to_end(device)),
                             ("parity", lambda: extra_parity(device))):
                try:
                    out[name] = fn()
                except Exception as exc:                                  # noqa: BLE001 — a secondary object must not cost the line
                    out[name] = {"error": f"{type(exc).__name__}: {exc}"}
            if isinstance(out.get("end_to_end"), dict) and "value" in out["end_to_end"]:
                out["value_end_to_end"] = out["end_to_end"]["value"]      # next to value and value_cold: what RunSimulation's call pattern sees
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    del scratch
    if rdv is not None:
        rdv.close()


if __name__ == "__main__":
    main()
