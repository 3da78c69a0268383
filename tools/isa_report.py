#!/usr/bin/env python3
"""Disassemble the gfx950 code object of a libsphmi.so build and report, for the neighbour kernels the bench launches
(or any kernel whose demangled name contains the given text): VGPRs / LDS / scratch from the code-object metadata, the
instruction mix of the whole kernel, and the pair loop (the innermost loop that holds the two `buffer_load_dwordx4`
gathers) instruction by instruction with --loop.

  python tools/isa_report.py [--lib PATH] [--match TEXT] [--loop] [--json OUT]
"""
import argparse, collections, json, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["k_neighbor_force<float, 3, 1, 33, 2, 2>", "k_neighbor_force<float, 3, 2, 33, 2, 2>"]


def code_object(lib, workdir):
    lib = os.path.abspath(lib)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], cwd=workdir, check=True, capture_output=True)
    for f in os.listdir(os.path.dirname(lib)):
        pass
    base = os.path.basename(lib)
    src_dir = os.path.dirname(os.path.abspath(lib))
    out = None
    for f in os.listdir(src_dir):
        if f.startswith(base + ".") and ("hipv4" in f or "host-x86_64" in f):
            p = os.path.join(src_dir, f)
            if "gfx950" in f:
                out = os.path.join(workdir, "dev.co"); os.replace(p, out)
            else:
                os.remove(p)
    if out is None:
        raise SystemExit("no gfx950 code object found in " + lib)
    return out


def metadata(co):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    info = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, None])[1]
        info[g("name")] = {"vgprs": int(g("vgpr_count")), "agprs": int(blk.split()[0]), "sgprs": int(g("sgpr_count")),
                           "lds_bytes": int(g("group_segment_fixed_size")), "scratch_bytes": int(g("private_segment_fixed_size")),
                           "max_flat_workgroup_size": int(g("max_flat_workgroup_size"))}
    return info


def kernels(co):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout.splitlines()
    out, cur = collections.OrderedDict(), None
    for ln in dis:
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            cur = m.group(1); out[cur] = []
        elif cur and ln.startswith("\t") and re.search(r"//\s*[0-9A-F]+:", ln):
            out[cur].append(ln)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, r))


def mnemonic(ln):
    return ln.strip().split()[0]


def pair_loop(lines):
    """The backward branch whose body holds exactly two buffer_load_dwordx4 and is shortest: [start, end) line indices."""
    addr = [int(re.search(r"//\s*([0-9A-F]+):", ln).group(1), 16) for ln in lines]
    best = None
    for k, ln in enumerate(lines):
        m = re.search(r"(s_cbranch_\w+|s_branch)\s+(\d+)\s", ln)
        if not m:
            continue
        off = int(m.group(2)); off = off - 65536 if off >= 32768 else off
        tgt = addr[k] + 4 + 4 * off
        if tgt >= addr[k]:
            continue
        j = next((i for i, a in enumerate(addr) if a == tgt), None)
        if j is None:
            continue
        body = lines[j:k + 1]
        if sum("buffer_load_dwordx4" in b for b in body) in (2, 4) and (best is None or len(body) < best[1] - best[0]):
            best = (j, k + 1)
    return best


def classify(mn):
    if mn.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")): return "trans"
    if mn.startswith("v_mfma"): return "mfma"
    if mn.startswith("v_"): return "valu"
    if mn.startswith(("buffer_", "global_", "flat_")): return "vmem"
    if mn.startswith("ds_"): return "lds"
    if mn.startswith("s_waitcnt"): return "waitcnt"
    if mn.startswith("s_nop"): return "nop"
    if mn.startswith("s_"): return "salu"
    return "other"


def report(lib, want, with_loop=False):
    """{demangled name: {symbol, isa_sha16, vgprs, lds_bytes, …, pair_loop}} for the kernels whose demangled name contains one of `want`."""
    import hashlib
    with tempfile.TemporaryDirectory() as d:
        co = code_object(lib, d)
        meta, ks = metadata(co), kernels(co)
    dm = demangle(list(ks))
    rep = {}
    for mangled, lines in ks.items():
        name = dm[mangled]
        if not any(w in name for w in want):
            continue
        mix = collections.Counter(classify(mnemonic(l)) for l in lines)
        text = "\n".join(re.sub(r"\s*//.*$", "", l.strip()) for l in lines)
        r = {"symbol": mangled, **meta.get(mangled, {}), "instructions": len(lines), "mix": dict(mix),
             "isa_sha16": hashlib.sha256(text.encode()).hexdigest()[:16]}
        pl = pair_loop(lines)
        if pl:
            body = lines[pl[0]:pl[1]]
            bm = collections.Counter(classify(mnemonic(l)) for l in body)
            r["pair_loop"] = {"instructions": len(body), "mix": dict(bm), "vector_alu_incl_trans": bm["valu"] + bm["trans"]}
            if with_loop:
                r["pair_loop"]["text"] = [re.sub(r"\s*//.*$", "", l.strip()) for l in body]
        rep[name] = r
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "sphexample_amd", "libsphmi.so"))
    ap.add_argument("--match", action="append")
    ap.add_argument("--loop", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    rep = report(a.lib, a.match or BENCH, with_loop=a.loop)
    for name, r in rep.items():
        print(f"{name}\n  isa {r['isa_sha16']} vgprs {r.get('vgprs')} agprs {r.get('agprs')} sgprs {r.get('sgprs')} lds {r.get('lds_bytes')} "
              f"scratch {r.get('scratch_bytes')} | {r['instructions']} instructions {r['mix']}")
        if "pair_loop" in r:
            print(f"  pair loop: {{k: v for k, v in r['pair_loop'].items() if k != 'text'}}".replace("{k: v for k, v in r['pair_loop'].items() if k != 'text'}", str({k: v for k, v in r['pair_loop'].items() if k != 'text'})))
            for l in r["pair_loop"].get("text", []):
                print("    " + l)
    if a.json:
        json.dump(rep, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
