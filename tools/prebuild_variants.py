#!/usr/bin/env python3
"""Build libsphmi variants HERE (hipcc cross-compiles without a GPU) into build/variants/ — git-ignored, but they travel to
the GPU box with the snapshot, so a sweep costs no GPU-minutes for compiling.
usage: python tools/prebuild_variants.py "name:-DSPHMI_X=1 -DSPHMI_Y=0" ...   (4 builds at a time)"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd import build  # noqa: E402
out_dir = os.path.join(ROOT, "build", "variants")
os.makedirs(out_dir, exist_ok=True)


def one(spec):
    name, _, flags = spec.partition(":")
    out = os.path.join(out_dir, f"libsphmi_{name}.so")
    build.build(force=True, extra_flags=flags.split(), out=out)
    open(out + ".flags", "w").write(flags + "\n")
    return name


with ThreadPoolExecutor(4) as ex:
    for n in ex.map(one, sys.argv[1:]):
        print("built", n, flush=True)
