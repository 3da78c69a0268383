"""In-tree build of libsphmi.so (hipcc, gfx950 only).  The built library is git-ignored but travels to
the GPU box with the repository snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libsphmi.so")
SOURCES = ["sphmi_engine.hip"]
HEADERS = ["sphmi_kernels.h", "sphmi_rebuild.h", "sphmi_multi.h", "sphmi_shm.h", os.path.join("..", "..", "include", "sphmi.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libsphmi.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """extra_flags / $SPHMI_CXXFLAGS carry experiment switches such as -DSPHMI_KSLOTS=16."""
    out = out or LIB
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("SPHMI_CXXFLAGS", "").split())
    if not force and not is_stale() and out == LIB and not extra_flags:
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           # the SLP vectoriser packs the distance tests into v_pk_*_f32 + v_mov shuffles (slower here)
           "-fno-slp-vectorize",
           "-mllvm", "-amdgpu-mfma-vgpr-form",   # MFMA results straight into VGPRs: no v_accvgpr_read per element
           *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


# Variant libraries the GPU test suite loads (built HERE by __graft_entry__.build(), git-ignored, shipped to the GPU box with
# the snapshot): every test that loads a library other than LIB loads one of these, so the driver can rebuild what it runs.
VARIANT_DIR = os.path.join(os.path.dirname(_HERE), "build", "variants")
VARIANTS = {
    # BASELINE config 3 names "LDS cell-tile staging on": the ablation build of profiles/HISTORY.md §4.5 (measured, off) — tests/test_lds_stage_variant_gpu.py
    "ldsstage": ("-DSPHMI_LDS_STAGE=1",),
}


def variant_path(name: str) -> str:
    return os.path.join(VARIANT_DIR, f"libsphmi_{name}.so")


def build_variants(force: bool = False, verbose: bool = False) -> dict:
    os.makedirs(VARIANT_DIR, exist_ok=True)
    out = {}
    for name, flags in VARIANTS.items():
        path = variant_path(name)
        fresh = os.path.exists(path) and all(os.path.getmtime(os.path.join(CSRC, f)) <= os.path.getmtime(path) for f in SOURCES + HEADERS)
        if force or not fresh:
            build(force=True, verbose=verbose, extra_flags=flags, out=path)
            with open(path + ".flags", "w") as f:
                f.write(" ".join(flags) + "\n")
        out[name] = path
    return out


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    if "--variants" in sys.argv:
        print(build_variants(force="--force" in sys.argv, verbose=True))
