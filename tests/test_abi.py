"""CPU-side checks of the drop-in boundary: the library builds/loads, exports every symbol
include/sphmi.h declares, the ctypes structs match the C structs, and the product path refuses to run
without a GPU instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from conftest import ROOT
from sphexample_amd._abi import ERR_ARGUMENT, ERR_DEVICE, SphmiConfig, SphmiError, SphmiProgress, make_config

HEADER = os.path.join(ROOT, "include", "sphmi.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sphmi_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sphexample_amd.engine import load_library
    lib = load_library()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libsphmi.so does not export {s}"


def test_struct_layout_matches_header():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "sphmi.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(sphmi_config), offsetof(sphmi_config, n_particles),
        offsetof(sphmi_config, rho0), offsetof(sphmi_config, eta2), sizeof(sphmi_progress), offsetof(sphmi_progress, delta_x)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    got = [C.sizeof(SphmiConfig), SphmiConfig.n_particles.offset, SphmiConfig.rho0.offset, SphmiConfig.eta2.offset,
           C.sizeof(SphmiProgress), SphmiProgress.delta_x.offset]
    assert [int(x) for x in out] == got


def test_backend_info_and_argument_errors(dam_break_2d):
    from sphexample_amd.engine import Engine, backend_info
    assert "gfx950" in backend_info() and "no CPU fallback" in backend_info()
    p, s = dam_break_2d
    cfg = make_config(len(p), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion)
    cfg.struct_size = 8
    with pytest.raises(SphmiError) as ei:
        Engine(cfg)
    assert ei.value.status == ERR_ARGUMENT


def test_no_cpu_fallback(dam_break_2d):
    """Without a GPU the product path must fail loudly (and with one, it must be the HIP path)."""
    import torch
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(SphmiError) as ei:
        make_engine(p, s)
    assert ei.value.status == ERR_DEVICE


def test_product_package_never_touches_the_oracle():
    """Nothing under sphexample_amd/ may import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "sphexample_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|libsphoracle|\borc_[a-z_]+\s*\(|oracle/|sph_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(text), f"{f} references the oracle"
