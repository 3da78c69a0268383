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
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert rf["peak_measured"] > 1000.0 and rf["launches"] > 0 and rf["avg_launch_ms"] > 0
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == j["unit"]


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
    assert ("falling back" in r.stderr) == (transport == "rccl")
