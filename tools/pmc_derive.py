#!/usr/bin/env python3
"""One output file of tools/pmc_passes.sh (raw per-dispatch averages, 8 counter groups) + the kernel_identity.json it wrote
→ the record bench.py quotes (`roofline.valu`, `roofline.traffic`) and refuses for any other kernel build.

  python tools/pmc_derive.py <passes.txt> <kernel_identity.json> <out.json> [comment]
"""
import json, os, re, sys
N = 1057738
BENCH = {"predictor": "k_neighbor_force<float, 3, 1, 33, 2, 2>", "corrector": "k_neighbor_force<float, 3, 2, 33, 2, 2>"}
if os.environ.get("PMC_KERNELS"):            # other kernels than the bench's: "predictor name|corrector name" ($PMC_FLOAT_BYTES = 8 for fp64 handles)
    BENCH = dict(zip(("predictor", "corrector"), os.environ["PMC_KERNELS"].split("|")))
FB = int(os.environ.get("PMC_FLOAT_BYTES", "4"))


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"## (.*?)\s+\(\d+ dispatches", line)
        if m:
            name = m.group(1)
            cur = next((k for k, v in BENCH.items() if v in name), None)
            if cur is None:
                cur = "k_init_reduce" if "k_init_reduce" in name else ("k_eos" if "k_eos" in name else None)
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)\s*$", line)
        if m and cur:
            out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
    return out


def derive(c):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0           # shader cycles of the launch (the counter sums the 8 XCDs)
    simd = 1024
    return {"launch_cycles": round(cyc), "valu_insts": c["SQ_INSTS_VALU"], "valu_insts_per_tile": round(c["SQ_INSTS_VALU"] / (N / 64.0)),
            "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (simd * cyc), 3),
            "cycles_per_valu_inst_issued": round(simd * cyc / c["SQ_INSTS_VALU"], 2),
            "waves_per_simd_mean": round(c["SQ_WAVE_CYCLES"] * 4 / (simd * cyc), 2),
            "wave_time_parked_on_waitcnt": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "wave_time_issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
            "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (simd * cyc), 3),
            "ta_busy_frac": round(c["TA_BUSY_avr"] / cyc, 3),
            "l1_hit_frac": round(1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 3),
            "l2_hit_frac": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in c else None,
            "salu_insts": c["SQ_INSTS_SALU"], "vmem_rd_insts": c["SQ_INSTS_VMEM_RD"], "lds_insts": c["SQ_INSTS_LDS"],
            "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
            "FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE")}


def main():
    passes, ident, out = sys.argv[1], sys.argv[2], sys.argv[3]
    comment = sys.argv[4] if len(sys.argv) > 4 else ""
    p = parse(passes)
    idj = json.load(open(ident))
    kernels = {}
    for role, name in BENCH.items():
        r = next(v for k, v in idj.items() if name in k)
        kernels[role] = {k: r[k] for k in ("symbol", "isa_sha16", "vgprs", "lds_bytes", "scratch_bytes", "instructions")}
        if "pair_loop" in r:
            kernels[role]["pair_loop_vector_alu"] = r["pair_loop"]["vector_alu_incl_trans"]
    counters = {role: derive(p[role]) for role in BENCH}
    rec = {"_comment": ("rocprofv3 --pmc passes of tools/pmc_passes.sh (8 counter groups, one run each, no tracing options) on `python bench.py "
                        "--steps 6 --warmup 2 --no-cpu-baseline --no-extras --precondition-ms 0` (3-D dam break, N = 1057738, fp32; 16 528 "
                        "tiles). Raw per-dispatch averages: " + os.path.basename(passes) + ". SQ_* ACTIVE / WAIT / WAVE_CYCLES count quad-cycles "
                        "(MI355X_MICROARCH.md); launch_cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs "
                        "x launch_cycles). `kernels` is the identity of the two kernels the counters were taken on (tools/isa_report.py): "
                        "bench.py quotes the counters only when the loaded library's kernels hash to the same ISA. " + comment),
           "n_particles": N, "kernels": kernels, "counters": counters}
    # traffic beyond the L2s: FETCH_SIZE (KiB) is calibrated IN THE SAME PASS on two kernels of known traffic, as
    # MI355X_MICROARCH.md prescribes for gfx950 (the counter reports half of a 16 B/lane read stream)
    cal = {}
    if "k_init_reduce" in p and "FETCH_SIZE" in p["k_init_reduce"]:
        known = 48.0 * (FB / 4) * N / 1024.0
        cal["k_init_reduce"] = {"known_read_KiB": round(known, 1), "FETCH_SIZE_KiB": p["k_init_reduce"]["FETCH_SIZE"], "ratio": round(p["k_init_reduce"]["FETCH_SIZE"] / known, 3)}
    if "k_eos" in p and "FETCH_SIZE" in p["k_eos"]:
        known = 32.0 * (FB / 4) * N / 1024.0
        cal["k_eos"] = {"known_read_KiB": round(known, 1), "FETCH_SIZE_KiB": p["k_eos"]["FETCH_SIZE"], "ratio": round(p["k_eos"]["FETCH_SIZE"] / known, 3),
                        "WRITE_SIZE_KiB": p["k_eos"].get("WRITE_SIZE")}
    if cal:
        ratio = sum(v["ratio"] for v in cal.values()) / len(cal)
        corr = 1.0 / ratio
        fetch = sum(p[r]["FETCH_SIZE"] for r in BENCH) / 2 * 1024.0
        write = sum(p[r]["WRITE_SIZE"] for r in BENCH) / 2 * 1024.0
        rec["traffic"] = {"calibration": cal, "fetch_correction": round(corr, 3),
                          "bytes_per_particle_per_launch_corrected": round((fetch * corr + write) / N, 1),
                          "bytes_per_particle_per_launch_uncorrected": round((fetch + write) / N, 1),
                          "algorithmic_bytes_per_particle_per_launch": 77.0 if FB == 4 else 153.0,
                          "note": "bytes leaving the L2s (Infinity-Cache hits included): fabric traffic, not HBM traffic — the three state sets fit the 256 MB Infinity Cache"}
    json.dump(rec, open(out, "w"), indent=1)
    keys = ("launch_cycles", "valu_insts_per_tile", "valu_busy_frac", "waves_per_simd_mean", "wave_time_parked_on_waitcnt", "ta_busy_frac", "l1_hit_frac",
            "mfma_busy_frac", "lds_insts", "lds_bank_conflict_cycles")
    for role, v in counters.items():
        print(f"{role:10s}", {a: v[a] for a in keys})
    print("traffic", rec.get("traffic", {}).get("bytes_per_particle_per_launch_corrected"))


if __name__ == "__main__":
    main()
