#!/usr/bin/env python3
"""profiles/r02_pmc_passes_*.txt (tools/pmc_passes.sh) → profiles/r02_valu_counters.json: the derived figures bench.py and
DESIGN.md quote.  usage: python tools/pmc_derive.py"""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1057738
def parse(path):
    out = {}; cur = None
    for line in open(path):
        m = re.match(r"## void sphmi::k_neighbor_force<float, 3, (\d)", line)
        if m: cur = {"1": "predictor", "2": "corrector"}[m.group(1)]; continue
        if line.startswith("###"): cur = "predictor"; continue     # (the pass header replaces the first kernel header)
        m = re.match(r"\s+(\w+)\s+([\d.]+)", line)
        if m and cur: out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
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
            "salu_insts": c["SQ_INSTS_SALU"], "vmem_rd_insts": c["SQ_INSTS_VMEM_RD"], "lds_insts": c["SQ_INSTS_LDS"],
            "lds_bank_conflict": c["SQ_LDS_BANK_CONFLICT"]}
rec = {"_comment": "rocprofv3 --pmc passes of tools/pmc_passes.sh (5 groups, one run each, no tracing options) on `python bench.py --steps 6 "
       "--warmup 2 --no-cpu-baseline` (3-D dam break, N = 1057738, fp32; 16 528 tiles = waves). Raw per-dispatch averages: "
       "profiles/r02_pmc_passes_*.txt. Derived by tools/pmc_derive.py: SQ_* ACTIVE / WAIT / WAVE_CYCLES count quad-cycles "
       "(MI355X_MICROARCH.md); launch_cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x "
       "launch_cycles); valu_insts_per_tile = wave instructions per 64 particles. `final` = the shipped kernel (interleaved records, "
       "4 tiles per block, queues of 12 entries drained by 1); `interleaved_queue8` = the same with round 1's queues (8 entries "
       "drained by 4); `trimmed_before_interleave` = the instruction-trimmed kernel on separate pk0 / pk1 arrays (16 segments per "
       "XCD, one tile per block); `round1_kernel` = the kernel round 1 shipped, on this round's boxes.", "n_particles": N}
for tag in ("final", "interleaved_queue8", "trimmed_before_interleave", "round1_kernel"):
    p = parse(os.path.join(ROOT, "profiles", f"r02_pmc_passes_{tag}.txt"))
    rec[tag] = {k: derive(v) for k, v in p.items()}
json.dump(rec, open(os.path.join(ROOT, "profiles", "r02_valu_counters.json"), "w"), indent=1)
keys = ("launch_cycles", "valu_insts_per_tile", "valu_busy_frac", "cycles_per_valu_inst_issued", "waves_per_simd_mean",
        "wave_time_parked_on_waitcnt", "ta_busy_frac", "l1_hit_frac", "mfma_busy_frac")
for tag in ("round1_kernel", "trimmed_before_interleave", "interleaved_queue8", "final"):
    for k, v in rec[tag].items():
        print(f"{tag:28s} {k:10s}", {a: v[a] for a in keys})
