"""bench.py keeps the driver's contract: one JSON line on stdout with the agreed keys, the roofline and (at N = 1) the CPU
baseline objects; without a GPU it refuses to run (no CPU path); under torchrun with more ranks than GPUs the ranks share
the device through the library's shared-memory transport and the line says so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--dp", "0.02",
                        "--precondition-ms", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert KEYS <= set(j) and "cpu_baseline" in j
    assert (j["n_gpus"], j["steps"], j["warmup"], j["dtype"], j["scaling"], j["higher_is_better"]) == (1, 4, 2, "f32", "weak", True)
    assert j["vs_baseline"] is None and j["value"] > 0 and "workload" in j["config"]
    rf = j["roofline"]
    # `bound` names the binding resource; achieved / peak / frac stay the HBM figure the metric asks for
    assert rf["bound"] == "gather_issue" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert j["value_excl_rebuild"] >= j["value"] and j["rebuild_ms_in_window"] >= 0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert rf["peak_measured"] > 1000.0 and rf["launches_expected"] == 2 * j["steps"] and rf["launch_time_samples"] > 0 and rf["avg_launch_ms"] > 0
    assert j["config"]["launch"] == "one process"
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == j["unit"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_line_of_rank_processes_over_the_rccl_branch(world):
    """The launch the driver uses for its scaling runs — `torch.distributed.run … bench.py --gpus N`, one rank process per slab, the slabs talking
    through the RCCL branch of the library — end to end on ONE GPU: librccl replaced by the checking double of tests/mock_rccl/ in its stream-ordered
    mode (the real library refuses ranks that share a device).  The line must name the RCCL transport and no fallback."""
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("more than one GPU: tests/test_multi_device_gpu.py runs this launch over the real RCCL")
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock_rccl"))
    try:
        import build as mock_build
        mock = mock_build.build()
    finally:
        sys.path.remove(os.path.join(ROOT, "tests", "mock_rccl"))
    env = dict(os.environ, SPHMI_TRANSPORT="rccl", SPHMI_RCCL_LIB=mock, MOCK_RCCL_ASYNC="1", MOCK_RCCL_ASYNC_DELAY_US="200", GPU_MAX_HW_QUEUES="24",
               MOCK_RCCL_TIMEOUT="60", SPHMI_BENCH_SETUP_TIMEOUT="200", SPHMI_BENCH_RUN_TIMEOUT="300")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(29551 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--dp", "0.012",
                        "--precondition-ms", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "VIOLATION" not in r.stderr and "RCCL set-up failed" not in r.stderr, r.stderr[-3000:]
    j = _line(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == world and j["scaling"] == "weak"
    par = j["config"]["parallelism"]
    assert "RCCL (ncclSend/ncclRecv" in par and "FALLBACK" not in par and "SHARED-MEMORY" not in par and "sphmi_create_rank" in par, par
    assert "librccl SUBSTITUTED" in par and "not a valid multi-GPU measurement" in par, par        # the line owns up to the stand-in
    assert j["value"] > 0 and j["steps"] == 6


@pytest.mark.gpu
@pytest.mark.parametrize("transport", [None, "rccl"])
def test_bench_line_two_ranks_on_one_gpu(transport):
    """transport = "rccl": RCCL is ASKED for and refuses (two ranks on one device) — every rank learns it, the run falls back to
    the shared-memory transport and still produces its line (what a node whose RCCL cannot be set up would get)."""
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("more than one GPU: the ranks would not share a device")
    env = dict(os.environ, SPHMI_SHM_TIMEOUT="60")
    if transport:
        env["SPHMI_TRANSPORT"] = transport
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547" if transport is None else "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--dp", "0.02",
                        "--precondition-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert KEYS <= set(j) and "cpu_baseline" not in j
    assert j["n_gpus"] == 2 and "SHARED-MEMORY" in j["config"]["parallelism"]
    assert ("RCCL set-up failed" in r.stderr) == (transport == "rccl")
    if transport == "rccl":                                 # the text names the slab and the call, not a bare "invalid usage"
        assert "slab" in r.stderr and "ncclCommInitRank" in r.stderr


def test_bench_without_a_launcher_spawns_its_ranks(monkeypatch):
    """`python bench.py --gpus N` started like the N = 1 line (no torch.distributed.run around it, so
    WORLD_SIZE is unset) used to leave with SystemExit and no line.  It now starts the N ranks itself — the contract's own command
    line — and falls into the one-process multi-device handle when they fail.  CPU part: the decision and the command."""
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPHMI_BENCH_LAUNCH"):
        monkeypatch.delenv(k, raising=False)
    calls = []
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(bench, "self_spawn", lambda world, argv: calls.append((world, list(argv))) or 0)
    bench.main()                                             # the ranks did the work: nothing else happens in this process
    assert calls == [(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])]
    # the ranks failed: this process goes on as --single-process (here, without a GPU, up to the device check)
    import torch
    if not torch.cuda.is_available():
        monkeypatch.setattr(bench, "self_spawn", lambda world, argv: 7)
        with pytest.raises(SystemExit, match="no CPU path"):
            bench.main()
    # under a launcher (WORLD_SIZE set) nothing is spawned
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "1")
    monkeypatch.setattr(bench, "self_spawn", lambda world, argv: (_ for _ in ()).throw(AssertionError("spawned under a launcher")))
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match="no CPU path"):
            bench.main()


def test_self_spawn_runs_the_contracts_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.self_spawn(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert "self-spawned" in seen["env"]["SPHMI_BENCH_LAUNCH"]


@pytest.mark.gpu
def test_bench_line_without_a_launcher():
    """The same on the device: `python bench.py --gpus 2 …` produces ONE line, and the line says how the ranks came to be.  On a
    one-GPU box the two ranks share the device (shared-memory transport, labelled); with two GPUs they run over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["SPHMI_SHM_TIMEOUT"] = "60"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--dp", "0.02",
                        "--precondition-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["steps"] == 4
    assert "self-spawned" in j["config"]["launch"] and "torch.distributed.run" in j["config"]["launch"]


def _identity():
    sys.path.insert(0, ROOT)
    import bench
    return bench.loaded_kernel_identity()


def test_counters_are_quoted_for_the_profiled_kernel_only(tmp_path, monkeypatch):
    """the PMC counters under profiles/ carry the identity of the kernels they were taken on (symbol,
    ISA hash, registers, LDS); bench.py refuses them — traffic / valu null, the reason on the line — for any other build."""
    sys.path.insert(0, ROOT)
    import bench
    ident = _identity()
    assert "error" not in ident, ident
    assert set(ident) == {"predictor", "corrector"} and all(len(v["isa_sha16"]) == 16 and v["vgprs"] > 0 for v in ident.values())
    counters = {k: {"valu_insts": 2.0e8, "valu_busy_frac": 0.8, "waves_per_simd_mean": 5.0, "wave_time_parked_on_waitcnt": 0.3} for k in ident}
    rec = {"n_particles": 1000000, "kernels": ident, "counters": counters, "traffic": {"bytes_per_particle_per_launch_corrected": 100.0}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    path = tmp_path / bench.COUNTER_RECORD
    json.dump(rec, open(path, "w"))
    traffic, valu, why = bench.counters_for(ident, 500000, 0.5)
    assert why is None and traffic == 5.0e7 and valu["wave_insts_per_launch"] == 1.0e8 and valu["busy_frac_pmc"] == 0.8
    other = json.loads(json.dumps(ident)); other["corrector"]["isa_sha16"] = "0" * 16
    traffic, valu, why = bench.counters_for(other, 500000, 0.5)
    assert traffic is None and valu is None and "another kernel" in why and "corrector isa_sha16" in why
    other = json.loads(json.dumps(ident)); other["predictor"]["vgprs"] += 8
    assert bench.counters_for(other, 500000, 0.5)[2].count("predictor vgprs") == 1
    os.remove(path)
    assert bench.counters_for(ident, 500000, 0.5)[:2] == (None, None)


def test_the_committed_counter_record_matches_the_tree():
    """profiles/r03_counters.json (when present) must describe the kernels of the library built from THIS tree — a kernel
    change without new counters shows up here, not as silently stale figures on the bench line."""
    sys.path.insert(0, ROOT)
    import bench
    path = os.path.join(ROOT, bench.COUNTER_RECORD)
    if not os.path.exists(path):
        pytest.skip("no counter record committed yet")
    ident = _identity()
    assert bench.counters_for(ident, 1000, 0.5)[2] is None


def test_no_neighbour_kernel_spills_or_calls_a_function(tmp_path):
    """Every instantiation of k_neighbor_force must be ONE straight kernel: no scratch (registers spilled to memory) and no device
    function left outside it.  Round 3 lost a factor five on `MovingSquare2d` (run-time models, 4 / 8 waves per tile) without a
    single failing test: once the pair loop had several bodies to choose from, the compiler stopped inlining the lambda that runs
    it — and a lambda that captures by reference and is CALLED keeps every accumulator of the tile in scratch (134 registers,
    976 bytes of scratch, 15 → 105 µs per launch).  The lambdas are force-inlined now; this reads the code object's metadata
    and symbol table (no GPU needed)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_report
    from sphexample_amd import build
    lib = build.build()
    co = isa_report.code_object(lib, str(tmp_path))
    meta = isa_report.metadata(co)
    names = isa_report.demangle(list(meta))
    assert sum("k_neighbor_force" in d for d in names.values()) > 100
    spilled = {names[k]: v["scratch_bytes"] for k, v in meta.items() if v["scratch_bytes"]}      # (true of every kernel of the library)
    assert not spilled, spilled
    outlined = [d for k, d in isa_report.demangle(list(isa_report.kernels(co))).items() if k not in meta]
    assert not outlined, outlined[:4]
