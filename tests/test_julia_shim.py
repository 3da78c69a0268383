"""Static guard of the Julia shim (julia/SPHExampleMI355X.jl) — the only reference-side artefact of the drop-in boundary.

The image has no Julia, so the shim has never executed; it is edited by hand whenever the ABI moves.  This test READS it:
  (a) `struct SphmiConfig` / `mutable struct SphmiProgress` equal `sphmi_config` / `sphmi_progress` of include/sphmi.h field for
      field — names, order, widths, array lengths;
  (b) every `ccall((:sym, LIB), Ret, (ArgTypes…), args…)` names a function the header declares, passes as many arguments as its
      type tuple lists, and the tuple equals the header's prototype in length and in class (pointer / 32-bit integer / 64-bit integer
      / double) argument for argument, return type included;
  (c) `ABI_VERSION` equals `SPHMI_ABI_VERSION`;
  (d) when /root/reference is present (this container, not the GPU box): every `SimKernel.x`, `SimConstants.x`, `SimMetaData.x`,
      `P.x` and MotionDetails field the shim touches is a field of the reference struct it binds
      (src/SPHKernels.jl:30-40, src/SimulationConstantsConfiguration.jl:36-52, src/SimulationMetaDataConfiguration.jl:28-67,
      src/PreProcess.jl:114, src/SimulationGeometry.jl:17-22), and the `SimulationLoop` method's argument list equals
      src/SPHCellList.jl:727-733 name for name.
A renamed header field, a dropped ccall argument or a reordered struct member fails here instead of on a user's machine."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "SPHExampleMI355X.jl")
HEADER = os.path.join(ROOT, "include", "sphmi.h")
REF = "/root/reference"


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _strip_jl_comments(text):
    return "\n".join(ln.split("#", 1)[0] if '"' not in ln.split("#", 1)[0] or ln.split("#", 1)[0].count('"') % 2 == 0 else ln for ln in text.splitlines())


def header_text():
    return _strip_c_comments(open(HEADER).read())


def shim_text():
    return _strip_jl_comments(open(SHIM).read())


C_WIDTH = {"int32_t": ("i", 4), "int64_t": ("i", 8), "uint64_t": ("i", 8), "double": ("f", 8), "uint8_t": ("i", 1), "int": ("i", 4)}
JL_WIDTH = {"Int32": ("i", 4), "Int64": ("i", 8), "UInt64": ("i", 8), "Float64": ("f", 8), "UInt8": ("i", 1), "Cint": ("i", 4)}


def c_struct_fields(name, macros):
    m = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), header_text(), flags=re.S)
    assert m, name
    out = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, rest = decl.split(None, 1)
        for item in rest.split(","):
            item = item.strip()
            am = re.match(r"(\w+)\s*\[\s*(\w+)\s*\]$", item)
            if am:
                n = am.group(2)
                out.append((am.group(1), C_WIDTH[ty], int(macros.get(n, n))))
            else:
                assert re.match(r"\w+$", item), decl
                out.append((item, C_WIDTH[ty], 1))
    return out


def jl_struct_fields(name):
    m = re.search(r"struct\s+%s\b(.*?)\nend" % name, shim_text(), flags=re.S)
    assert m, name
    out = []
    for item in re.split(r"[;\n]", m.group(1)):
        item = item.strip()
        fm = re.match(r"(\w+)::(.+)$", item)
        if not fm:
            continue                      # the inner constructor line of SphmiProgress
        ty = fm.group(2).strip()
        nt = re.match(r"NTuple\{\s*(\d+)\s*,\s*(\w+)\s*\}$", ty)
        out.append((fm.group(1), JL_WIDTH[nt.group(2)], int(nt.group(1))) if nt else (fm.group(1), JL_WIDTH[ty], 1))
    return out


def macros():
    return dict(re.findall(r"#define\s+(SPHMI_\w+)\s+(\d+)", header_text()))


def test_config_and_progress_structs_match_the_header():
    mc = macros()
    for jl, c in (("SphmiConfig", "sphmi_config"), ("SphmiProgress", "sphmi_progress")):
        a, b = jl_struct_fields(jl), c_struct_fields(c, mc)
        assert [f[0] for f in a] == [f[0] for f in b], (jl, "field names / order")
        assert a == b, (jl, [(x, y) for x, y in zip(a, b) if x != y])


def test_abi_version_matches():
    m = re.search(r"const\s+ABI_VERSION\s*=\s*Int32\((\d+)\)", shim_text())
    assert m and m.group(1) == macros()["SPHMI_ABI_VERSION"]


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def shim_ccalls():
    text = shim_text()
    calls = []
    for m in re.finditer(r"ccall\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        parts = _split_top(text[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*LIB\s*\)$", parts[0])
        assert sym, parts[0]
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), parts
        calls.append((sym.group(1), parts[1], _split_top(types[1:-1].rstrip(",")), parts[3:]))
    return calls


def c_prototypes():
    protos = {}
    text = header_text()
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    for m in re.finditer(r"(const\s+char\s*\*|int32_t|int)\s+(sphmi_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",")] if m.group(3).strip() not in ("", "void") else []
        protos[m.group(2)] = (m.group(1), args)
    return protos


def c_class(decl):
    decl = decl.strip()
    if "*" in decl:
        return "ptr"
    ty = decl.replace("const", " ").split()[0]
    kind, width = C_WIDTH[ty]
    return f"{kind}{width}"


def jl_class(ty):
    ty = ty.strip()
    if ty.startswith(("Ptr{", "Ref{")) or ty == "Cstring":
        return "ptr"
    kind, width = JL_WIDTH[ty]
    return f"{kind}{width}"


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    calls = shim_ccalls()
    assert len(calls) >= 14 and {"sphmi_create", "sphmi_upload", "sphmi_advance", "sphmi_download_begin", "sphmi_download_end",
                                 "sphmi_download_permutation", "sphmi_destroy", "sphmi_last_error"} <= {c[0] for c in calls}
    for sym, ret, types, args in calls:
        assert sym in protos, f"{sym}: not declared in include/sphmi.h"
        c_ret, c_args = protos[sym]
        assert len(types) == len(args), f"{sym}: {len(types)} argument types, {len(args)} arguments passed"
        assert len(types) == len(c_args), f"{sym}: the shim passes {len(types)} arguments, the header declares {len(c_args)}"
        assert jl_class(ret) == ("ptr" if "*" in c_ret else "i4"), f"{sym}: return type {ret} against `{c_ret}`"
        for k, (jt, cd) in enumerate(zip(types, c_args)):
            assert jl_class(jt) == c_class(cd), f"{sym}: argument {k + 1} is {jt} in the shim and `{cd}` in the header"


def _ref_struct_fields(path, struct_name):
    text = open(os.path.join(REF, path), encoding="utf-8").read()
    m = re.search(r"struct\s+%s\b.*?\n(.*?)\nend" % struct_name, text, flags=re.S)
    assert m, (path, struct_name)
    return set(re.findall(r"^\s*([^\W\d][\w⁻¹²₀ᵩ]*)\s*::", m.group(1), flags=re.M))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine")
def test_the_shim_binds_fields_the_reference_has():
    text = shim_text()
    ident = r"([^\W\d][\w⁻¹²₀ᵩ⁺]*)"
    used = lambda var: set(re.findall(r"\b%s\.%s" % (var, ident), text))  # noqa: E731
    kern = _ref_struct_fields("src/SPHKernels.jl", "SPHKernelInstance")
    assert used("SimKernel") and used("SimKernel") <= kern, used("SimKernel") - kern
    cons = _ref_struct_fields("src/SimulationConstantsConfiguration.jl", "SimulationConstants")
    assert len(used("SimConstants")) >= 11 and used("SimConstants") <= cons, used("SimConstants") - cons
    meta = _ref_struct_fields("src/SimulationMetaDataConfiguration.jl", "SimulationMetaData")
    assert used("SimMetaData") and used("SimMetaData") <= meta, used("SimMetaData") - meta
    motion = _ref_struct_fields("src/SimulationGeometry.jl", "MotionDetails")
    assert used("m") == {"Direction", "Velocity", "StartTime", "Duration"} and used("m") <= motion, used("m") - motion
    pre = open(os.path.join(REF, "src/PreProcess.jl"), encoding="utf-8").read()
    sa = re.search(r"SimParticles\s*=\s*StructArray\(\((.*?)\)\)", pre, flags=re.S)
    columns = set(re.findall(r"(\w+)\s*=", sa.group(1)))
    assert len(columns) == 17
    assert used("P") and used("P") <= columns, used("P") - columns
    # the CubicSpline tensile parameter the shim forwards (src/SPHKernels.jl:15-19)
    assert "eps" in _ref_struct_fields("src/SPHKernels.jl", "CubicSpline")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine")
def test_simulationloop_signature_is_the_references():
    def arg_names(text, start):
        i = text.index("(", start) + 1
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        return [re.match(r"\s*([^\s:]+)", a).group(1) for a in _split_top(text[i:j - 1])]
    ref = open(os.path.join(REF, "src/SPHCellList.jl"), encoding="utf-8").read()
    r0 = ref.index("function SimulationLoop(")
    shim = open(SHIM, encoding="utf-8").read()
    s0 = shim.index("function SimulationLoop(")
    a, b = arg_names(ref, r0), arg_names(shim, s0)
    assert len(a) == 19 and a == b, list(zip(a, b))
    # and it extends the reference's function, not a namesake
    assert re.search(r"import\s+SPHExample\.SPHCellList:\s*SimulationLoop", shim)
