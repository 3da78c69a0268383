"""Build of the RCCL test double (tests/mock_rccl/mock_rccl.cpp -> tests/mock_rccl/libmockrccl.so; hipcc, host code only).
Test infrastructure: __graft_entry__.build() calls this so that the library travels to the GPU box with the snapshot;
nothing under sphexample_amd/ knows it exists — the tests hand its path to libsphmi through $SPHMI_RCCL_LIB."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mock_rccl.cpp")
LIB = os.path.join(HERE, "libmockrccl.so")


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", SRC, "-o", LIB, "-lrt", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
