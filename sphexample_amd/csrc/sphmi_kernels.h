// sphmi_kernels.h — gfx950 (CDNA4, wave64) device kernels of the SPH neighbour + force engine.
//
// Data layout in HBM (SoA of 16-/32-byte packets, cell-sorted, x-fastest cell order):
//   pk0[i] = { x, y, z, ρ·s }   s = +1 for Fluid (MotionLimiter = 1), −1 otherwise
//   pk1[i] = { vx, vy, vz, P }  (state set "A"/"B")   or   { v⁺, ρⁿ·s } (half-step set "H")
// 2-D runs keep z = vz = 0 and use component D−1 as the gravity / hydrostatic axis.
//
// k_neighbor_force replaces NeighborLoop! + ComputeInteractions! + ReductionStep! + HalfTimeStep /
// FullTimeStep + LimitDensityAtBoundary! + DensityEpsi! + Pressure! + the Δt / Δx reductions of the
// reference (src/SPHCellList.jl:168-217, 268-317, 367-381, 624-652, 706-724;
// src/SimulationEquations.jl:9-42; src/TimeStepping.jl:24-46) with ONE launch per pass.
//
// Mapping (one TILE = 64 consecutive sorted target particles; which tile a block takes comes from the tile schedule built with
// the cell list — sphmi_rebuild.h).  The kernels of two, four and eight waves per tile — every default launch since the second half
// of round 4 — serve a tile as two HALF TILES: a wave holds 32 targets with TWO LANES PER TARGET (lanes l and l + 32), which take
// alternate groups of four candidates of every chunk the wave scans and add their sums after the loop; a half is worked off by one,
// two or four waves that deal its chunks in turn (`kHalf`, profiles/HISTORY.md §4.8).  One wave per tile with one lane per target is the
// organisation of rounds 1–4 ($SPHMI_WPT=1; the description below is written for it, the phases are the same):
//   phase 1  "who is within H".  For each of the 3^(D-1) cell rows around the tile the three x-adjacent
//            cells of every target are one contiguous particle range (x is the fastest sort axis); the
//            union over the tile is scanned in 64-candidate chunks, one candidate per lane, loaded
//            coalesced.  The 64×64 matrix |c−t|² − H'² of a chunk comes from the matrix cores in tile-local coordinates,
//            for fp32 AND fp64 handles (the mask only has to be a superset; the pair loop is exact) — half tiles: ONE
//            v_mfma_f32_32x32x16_f16 per 32×32 block on hi + lo split operands (the f32-input instruction runs on the
//            vector ALU's multipliers and stalls every wave of the SIMD; scan_chunk16), full tiles: three
//            v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain); every lane ends up with the results
//            of ITS target, packs their sign bits with v_alignbit, clears the bits outside its own three cells
//            and pushes each non-empty 32-candidate half as { mask, index of bit 0 } onto its private LDS queue.
//   phase 2  "pair physics".  A lane fetches its next non-empty mask the moment the current one is used up,
//            walks the set bits, gathers the two neighbour packets through buffer loads and accumulates the
//            pair terms — no intra-wave synchronisation, no iterations on empty masks, no atomics, each output
//            written once.  Phase 1 resumes when some lane's queue is full.
//   epilogue predictor or corrector fused in; wave-level max-reductions for Δt / Δx (consumed on the device by
//            k_step_control).
// What ships: DESIGN.md §4.  Measured history and the experiments behind these choices: profiles/HISTORY.md §4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sphmi {

template <class T> struct Vec4;
template <> struct Vec4<float>  { using type = float4;  };
template <> struct Vec4<double> { using type = double4; };

// One half of the interleaved particle records.  A particle's two packets — pk0 = {x, y, z, ρ·s}, pk1 = {v, P} —
// sit side by side in ONE array, rec[2i] and rec[2i + 1]: the two gathers of a neighbour then touch ONE cache line
// instead of one line in each of two arrays (−4 % kernel time at 1.06 M particles, measured by pointing the second
// gather at the line of the first).  Half<V>{rec + h}[i] is packet h of particle i; kernels index it like an array.
template <class V> struct Half {
    V* p;
    __host__ __device__ Half() : p(nullptr) {}
    __host__ __device__ explicit Half(V* q) : p(q) {}
    template <class U> __host__ __device__ Half(const Half<U>& o) : p(o.p) {}           // Half<V4> → Half<const V4>
    __host__ __device__ V& operator[](long long i) const { return p[2 * i]; }
    __host__ __device__ explicit operator bool() const { return p != nullptr; }
};

constexpr int kWave = 64;
// Build switches that remain.  Everything else that rounds 2 and 3 measured and left off (role bits in the queue entries, half
// prefetch, two neighbours in flight, phase 1 pipelined, s_setprio, the predictor's masks handed to the corrector, the sign bits
// on the matrix pipe; round 5: packed fp32 arithmetic, lane pairs that gather one record per instruction) lives as patches under
// profiles/ (r03_raw/mask_mfma_experiment.patch, r04_retired_switches.patch, r05_retired_switches.patch,
// r05_pair_gather_experiment.patch, r05_deep_pair_experiment.patch) with its figures in profiles/r03_pair_loop_experiments.md and profiles/HISTORY.md §4.9 — not in this file.
#ifndef SPHMI_LDS_STAGE
#define SPHMI_LDS_STAGE 0       // ABLATION BUILD (BASELINE config 3: "LDS cell-tile staging on"): the candidate records of a chunk are staged in LDS
                                // and the pair loop reads them from there, chunk by chunk, instead of gathering from L1 through per-lane mask queues
                                // (build/variants/libsphmi_ldsstage.so: __graft_entry__.build(); parity: tests/test_lds_stage_variant_gpu.py)
#endif
#ifndef SPHMI_HALF_TILE
#define SPHMI_HALF_TILE 1       // two-wave tiles: 1 = each wave serves 32 targets, two lanes per target (k_neighbor_force, kHalf); 0 = both waves serve
                                // the tile's 64 targets and split its chunks (rounds 1-4) — A/B builds only
#endif
#ifndef SPHMI_HALF4
#define SPHMI_HALF4 1           // four-wave tiles: 1 = two half tiles of two waves each (the waves of a half deal its chunks alternately), 0 = four waves
                                // that serve the tile's 64 targets and split its chunks (rounds 2-4) — A/B builds only
#endif
#ifndef SPHMI_BALANCE
#define SPHMI_BALANCE 1         // interleaved half tiles: per chunk the lane of a target that has had FEWER pairs so far takes the larger of the two interleaved shares
                                // (whole 32-bit words change hands; see the push of phase 1).  0 = every lane keeps the share the matrix layout hands it (A/B builds)
#endif
#ifndef SPHMI_PAIR_FETCH64
#define SPHMI_PAIR_FETCH64 1    // fp64 half tiles of one wave per half, compiled-in models: adjacent lanes fetch the two 16-byte halves of ONE packet per instruction
                                // (run_pairs_piped, kPairFetch64).  0 = four gathers of a lane's own record (A/B builds; what every other kernel does)
#endif
#ifndef SPHMI_F16_SCAN
#define SPHMI_F16_SCAN 1        // half tiles: the distance matrix of phase 1 from ONE v_mfma_f32_32x32x16_f16 per 32x32 block instead of three
                                // v_mfma_f32_32x32x2_f32 (see scan_chunk16).  0 = the f32-input form (A/B builds; what full tiles keep)
#endif
#ifndef SPHMI_HALF_INTERLEAVE
#define SPHMI_HALF_INTERLEAVE 1 // half tiles: the two lanes of a target take alternate groups of FOUR candidates (1) or the lower / upper 32 of a chunk (0: A/B builds).
                                // (Round 5 built alternate SINGLE candidates — no index arithmetic per pair — and packed fp32 arithmetic for the pair; both measured
                                // slower and live in profiles/r05_retired_switches.patch, profiles/HISTORY.md §4.9.)
#endif
#ifndef SPHMI_SMALL_TRIMS
#define SPHMI_SMALL_TRIMS 1     // launches of four and eight waves per tile (a few hundred waves, 10 µs): the epilogue's loads requested at the wave's start, no
                                // pre-test in front of the reduction atomics (round 5); 0: A/B builds
#endif
#ifndef SPHMI_PREFETCH_RANGES
#define SPHMI_PREFETCH_RANGES 1 // launches of four and eight waves per tile: the cell ranges of all rows requested at once, parked in a private LDS column (round 6; 0: one exposed look-up per row)
#endif
#ifndef SPHMI_LOOP_UNROLL
#define SPHMI_LOOP_UNROLL 2     // fp32 pair loop (one pair per iteration): iterations per loop test (1: rounds 1-4)
#endif
#ifndef SPHMI_FAST_PAIR
#define SPHMI_FAST_PAIR 1       // fp32 kernels of the compiled-in models: pair_fast (one reciprocal per pair, transcendentals back to back, lane constants out of
                                // the loop — round 5); 0 = pair_core for every kernel (rounds 1-4), A/B builds
#endif
#ifndef SPHMI_DIAG
#define SPHMI_DIAG 0            // 1 / 2 / 4 / 5 / 8: diagnostic builds with WRONG results — gathers without arithmetic / arithmetic without gathers /
                                // neither / adjacent lanes sharing a gathered record (profiles/HISTORY.md §4.6) / adjacent lanes walking the union of their masks (§4.9)
                                // 16 (round 6): the PRICE of an LDS record ring under the per-lane queues — the pair loop of the fp32 half-tile kernels of one wave per half
                                // reads both packets of a neighbour with two ds_read_b128 from a ring in LDS, at the address its record index maps to, instead of gathering
                                // them through the texture path.  SPHMI_RING_RECORDS (a power of two) records of 32 bytes per WAVE, or per four-wave BLOCK with
                                // SPHMI_RING_SHARED=1 — the LDS allocation, and with it the occupancy, of the real design; SPHMI_RING_STAGE=1 adds what filling the ring costs
                                // (the second packet of every scanned chunk loaded with the first, two ds_write_b128 per lane and chunk; shared ring: by one wave in four).
                                // Nothing guarantees that the slot still holds the record: wrong results.  profiles/r06_lds_ring_price.md
#endif
#ifndef SPHMI_RING_RECORDS
#define SPHMI_RING_RECORDS 128
#endif
#ifndef SPHMI_RING_SHARED
#define SPHMI_RING_SHARED 0
#endif
#ifndef SPHMI_RING_STAGE
#define SPHMI_RING_STAGE 0
#endif
#ifndef SPHMI_RING_SOA
#define SPHMI_RING_SOA 0        // the ring as two arrays of packets (16-byte stride: sixteen bank positions for the sixteen lanes of a ds_read_b128 group) instead of
                                // records (32-byte stride: eight)
#endif

// Queue depth of the per-lane queues of non-empty 32-candidate accept masks.  A simulation of the queues on real tiles (profiles/HISTORY.md §4.4)
// and the loop counters of the kernel agree: with 8 entries drained by 4 (round 1) the lanes of a wave are busy in 70–72 % of the
// pair-loop iterations (the bound set by the lane with the most neighbours is 88 %); 8 drained by 1: 75 %; 10: 82 %; 12: 87 %; 16: 88 %.
// LDS pays for the depth — 160 KB per compute unit over the resident waves.  Measured at 1.06 M / 2.85 M particles (updates/s, fp32):
// 8 → 1.013e9 / –, 10 (predictor) + 11 (corrector) → 1.050e9 / 1.074e9, 12 → 1.050–1.060e9 / 1.097e9, 13 → 1.052e9 / 1.089e9,
// 14 → 1.037e9, 16 → 0.983e9 / 1.019e9: 12 entries = 6 KB per wave = six four-tile blocks (24 waves) per compute unit.
// The fp64 kernels hold ≤ 16 waves per unit by their registers: 16 entries.
// Eight waves per tile (the smallest cases; four waves per half tile) scan every fourth chunk of their half: eight entries.
#ifndef SPHMI_QCAP_HALF2
#define SPHMI_QCAP_HALF2 6      // fp32 half tiles of ONE wave per half, compiled-in models (the launches of 3 000 tiles and more: the bench)
#endif
#ifndef SPHMI_QCAP_HALF4
#define SPHMI_QCAP_HALF4 6       // … and of TWO waves per half (330 … 3 000 tiles: 70 k / 102 k / 159 k particles −3.3 / −1.9 / −1.3 % per step against twelve; profiles/r05_raw/qcap4_ab.txt)
#endif
template <class T, int WPT = 1, int MODEL = 0> constexpr int queue_entries() {
    // (fp64 half tiles: 10 / 12 / 16 / 20 entries all within 0.5 % of each other at 1.06 M / 159 k / 70 k particles: sixteen stay)
    // (half tiles of two / four waves per half: 8 / 10 / 12 / 16 and 6 / 8 / 10 / 12 entries within 1 % of each other from 273 to 2 482 tiles)
    // (round 5, after the two lanes of a target had learnt to share evenly (SPHMI_BALANCE): fp32 half tiles of one wave per half with a compiled-in model run
    // 4 / 5 / 6 / 10 entries at 0.4161 / 0.4081 / 0.4093 / 0.4160 ms per launch at C3, five interleaved repetitions — more, shorter bursts of the pair loop,
    // whose lanes drift less apart: 8.15 instead of 7.03 M gathers per launch at 30 instead of 36 CU-cycles each, the L1 hits 81-83 % instead of 77;
    // it is the corrector's queue that matters (predictor 6 + corrector 10: no gain).  The run-time models lose 5-12 % with six and keep ten,
    // every other class is within noise of its value: profiles/r05_raw/qcap_*.txt)
    return WPT >= 8 ? 8 : (sizeof(T) == 8 ? 16 : (WPT == 2 && SPHMI_HALF_TILE != 0 && SPHMI_LDS_STAGE == 0 ? (MODEL >= 0 ? SPHMI_QCAP_HALF2 : 10) : (WPT == 4 && MODEL >= 0 && SPHMI_HALF4 != 0 && SPHMI_HALF_TILE != 0 && SPHMI_LDS_STAGE == 0 ? SPHMI_QCAP_HALF4 : 12)));
}
// a full queue is consumed down to QUEUE − 1 − slack entries before scanning goes on: one, and two for the six-entry queues (0.4228 → 0.4192 ms per launch at C3,
// six interleaved repetitions; applied to every class it moves the 82 instantiations by −0.8 % in the geometric mean and single ones by ±1.4 %: profiles/r05_raw/slack_*.txt)
template <class T, int WPT = 1, int MODEL = 0> constexpr int queue_slack() { return queue_entries<T, WPT, MODEL>() == SPHMI_QCAP_HALF2 && SPHMI_QCAP_HALF2 < 8 ? 2 : 1; }

enum { PASS_FORCES_ONLY = 0, PASS_PREDICTOR = 1, PASS_CORRECTOR = 2 };
// model tags (values of include/sphmi.h)
enum { kViscZero = 0, kViscArtificial = 1, kViscLaminar = 2, kViscLaminarSPS = 3 };
enum { kDdtNone = 0, kDdtZeroGravityLinear = 1, kDdtLinear = 2, kDdtComplex = 3 };
constexpr int kModelDefault = kViscArtificial | (kDdtLinear << 4), kModelGeneric = -1;
// bit 8 of a compiled-in model tag: the kernel is cut off BEFORE it vanishes (H = k·h with k < 2 — example/DucklingMDBC.jl: 1.5), so the
// r² ≤ H² test of src/SPHCellList.jl:275 is applied per pair.  Round 4: DucklingMDBC ran the run-time variant for this flag alone.
constexpr int kModelCutBit = 256, kModelDefaultCut = kModelDefault | kModelCutBit;

// Device-side step control (Engine::advance): everything the while loop of src/SPHCellList.jl:742-802 decides per
// step lives here, so the host can queue several steps without a round trip and look at the flags afterwards.
struct StepCtrl {
    double dt, dt2;            // Δt of the step being executed
    double delta_x;            // Δx accumulator of update_delta_x! (:706-724)
    double total_time;         // SimMetaData.TotalTime
    double t_step_start;       // TotalTime at the start of the current step (ProgressMotion's clock)
    double t_target;           // loop bound: `while TotalTime <= t_target`
    double last_visc, last_amax;
    double last_dt;            // Δt of the last step that was actually executed (SimMetaData.CurrentTimeStep)
    long long steps_done, max_steps;
    int active;                // 1: the kernels of this step run; 0: they return at once
    int need_rebuild;          // Δx ≥ h: the host rebuilds the cell list, clears the flag and re-queues the step
    int resume;                // the step after a rebuild re-uses the Δt already computed
    int stop;                  // TotalTime > t_target or max_steps reached
    int error;                 // 1: non-positive / NaN Δt, 2: non-positive density, 3: a particle left the cell grid of a device-side rebuild, 4: a peer slab never posted (mailbox exchange)
    int pre_rebuilt;           // the rebuild this control is about to ask for has been served already: every SimulationLoop call opens
                               // with one (Δx re-armed to 1 + h, src/SPHCellList.jl:739,758-762), so the host runs it BEFORE queueing the
                               // first step instead of queueing a one-step batch for the control to cancel (≈60 µs per call)
};

// one thread: the per-step decisions of Δt (src/TimeStepping.jl:30-43) and update_delta_x! on the reduction slots the
// previous corrector filled (bit patterns of non-negative values)
// The decisions of one step on the four reduction slots the previous corrector filled (r0 … r3: bit patterns of
// non-negative values) and the control block `c` (a copy; the caller stores it).  Returns true when the slots were consumed
// (the step runs: they must be zero before its corrector fills them again), false when they must survive — a step that
// ends the interval, asks for a rebuild or finds an error leaves them for the control that follows.
template <class T>
__device__ __forceinline__ bool step_control_decide(unsigned long long r0, unsigned long long r1, unsigned long long r2,
                                                    unsigned long long r3, StepCtrl& c, double h, double c0, double CFL) {
    if (c.stop || c.error || c.need_rebuild) { c.active = 0; return false; }
    if (!c.resume) {
        // a non-positive density of the LAST corrector is an error of this call even when the loop bound ends it here: the state
        // set it went into must not be handed out as if it were good (found by tests/test_fuzz_gpu.py: the flag used to be
        // looked at after the loop bound, so a bad last step of an interval surfaced one sphmi_advance late)
        if (r3) { c.error = 2; c.active = 0; return false; }
        if (!(c.total_time <= c.t_target) || (c.max_steps >= 0 && c.steps_done >= c.max_steps)) { c.stop = 1; c.active = 0; return false; }
        auto dec = [](unsigned long long b) -> double {
            if constexpr (sizeof(T) == 4) return (double)__uint_as_float((unsigned)b); else return __longlong_as_double((long long)b);
        };
        const double maxdisp = sqrt(dec(r0)), visc = dec(r1), amax = sqrt(dec(r2));
        c.delta_x += 4.0 * maxdisp;
        const double dt1 = sqrt(h / amax), dt2 = h / (c0 + visc);
        const double dt = CFL * (dt1 < dt2 ? dt1 : dt2);
        c.last_visc = visc; c.last_amax = amax;
        c.dt = dt; c.dt2 = dt * 0.5;
        if (!(dt > 0.0) || dt != dt || c.delta_x != c.delta_x) { c.error = 1; c.active = 0; return false; }
        if (c.delta_x >= h) {
            if (c.pre_rebuilt) { c.pre_rebuilt = 0; c.delta_x = 0.0; }          // :760-761, served before it was asked for
            else { c.need_rebuild = 1; c.resume = 1; c.active = 0; return false; }
        }
    }
    c.resume = 0;
    c.t_step_start = c.total_time;
    c.total_time += c.dt;                                   // UpdateMetaData!, :679-685 (nothing reads it before the
    c.steps_done += 1;                                      // next control except through t_step_start)
    c.last_dt = c.dt;
    c.active = 1;
    return true;
}

// one thread: the control as a launch of its own (handles with mDBC or moving bodies, whose kernels need the decisions before
// the predictor; slab handles, whose slots are MAX-allreduced first).  State and slots are read in one burst.
template <class T>
__global__ void k_step_control(unsigned long long* red, StepCtrl* cp, double h, double c0, double CFL) {
    StepCtrl c = *cp;
    const unsigned long long r0 = red[0], r1 = red[1], r2 = red[2], r3 = red[3];
    const bool consumed = step_control_decide<T>(r0, r1, r2, r3, c, h, c0, CFL);
    *cp = c;
    if (consumed) { red[0] = 0; red[1] = 0; red[2] = 0; red[3] = 0; }
}

template <class T>
struct ForceParams {
    using V4 = typename Vec4<T>::type;
    Half<const V4> src0; // neighbour stream packet 0 (A for pass 1, H for pass 2) — src1.p == src0.p + 1: one record array
    Half<const V4> src1; // neighbour stream packet 1
    Half<const V4> a0;   // state A (corrector epilogue)
    Half<const V4> a1;
    Half<V4> out0;       // H (predictor) or B (corrector)
    Half<V4> out1;
    V4* accbuf;          // { a, dρ/dt }
    // fp32 handles: the low words of the double-float state, { x_lo, y_lo, z_lo, ρ_lo } per particle (state A / B: x = pk0.xyz + lo,
    // ρ = |pk0.w| + lo).  The record sets hold what the NEIGHBOURS see — the state rounded to fp32, an error of half an ulp that
    // does not accumulate; the corrector epilogue integrates the full value, so that the state no longer loses one fp32 rounding of
    // ρ ≈ 1000 and of x per step (round 3: ρ error 5.9e-6 after 100 steps at 1 M particles, linear in the step count).  In place:
    // only the lane of a particle ever touches its entry.  Null: fp64 handles (and $SPHMI_COMPENSATE=0).
    V4* comp;
    const int* key;      // padded linear cell id of every sorted particle
    const int* cstart;   // exclusive scan of cell counts, ncell+1 entries
    const uint8_t* type;
    unsigned long long* red;   // [0] max |x⁺−x|², [1] max visc, [2] max |a|² (bit patterns), [3] bad-ρ flag
    unsigned long long* stats; // loop counters of -DSPHMI_STATS builds
    const StepCtrl* ctrl;      // device-side step control (null: dt / dt2 below are used, the kernel always runs)
    // Control taken INSIDE the predictor (plain handles: no mDBC, no moving body, no slab): every block takes the decisions
    // of the step itself from the state the previous step left (`ctl_in`) and the slots its corrector filled (`red_in`) —
    // the same few hundred instructions on the same inputs in every block, instead of a one-thread launch in front of every
    // step (4.4 of the 32.5 µs of a 2-D dam-break step).  Block 0 stores the decided state to `ctl_out` — the OTHER
    // control block: late blocks of this launch still read `ctl_in` — where this step's corrector and the next step's
    // predictor read it, and zeroes `red_zero`, the OTHER set of slots, which this step's corrector fills.  Null: `ctrl`.
    const StepCtrl* ctl_in; StepCtrl* ctl_out;
    const unsigned long long* red_in; unsigned long long* red_zero;
    double ctl_h, ctl_c0, ctl_CFL;
    // mDBC handles take the control inside k_mdbc (MdbcParams::ctl_in), which runs before this kernel: the predictor of an
    // executed step then only zeroes what k_mdbc could not zero itself — slots 0–2 of the set this step's corrector fills
    // (mdbc_zero[0..2]; slot 3 holds k_mdbc's flag of this step) and the flag slot of the other set (mdbc_flag_zero), which the
    // next step's k_mdbc and corrector fill.  Null: nothing to do.
    unsigned long long* mdbc_zero; unsigned long long* mdbc_flag_zero;
    const int* order;    // tile schedule: block b of XCD run x = b % 8 processes tile order[part[x] + b / 8]
    const int* part;     // [0..7] first entry of run x in order[], [8..15] tiles in run x
    int* tile_work;              // sampled launch: tile_work[tile] = 9·pair-loop iterations + 16·chunks of this tile (of its slowest wave × WPT), or null
    unsigned long long* xcd_clock;   // sampled launch: [x] = latest tile end on XCD x (max), [8] = earliest tile start (min); else null
    unsigned long long* trace;   // experiment builds (SPHMI_STATS / SPHMI_TRACE): per tile { start, end } of s_memrealtime, or null
    int N, nxp, nxyp;
    int visc, ddt, shift;    // model tags for the run-time variant of the kernel
    int exact_cut;           // H < 2h: apply r² ≤ H² per pair (the compiled-in variant requires H = 2h)
    int kernel;              // 0 WendlandC2, 1 CubicSpline (run-time variant only)
    V4* kout;                // StoreKernelOutput: { Σ∇W, ΣW } of the corrector pass, or null
    T dt, dt2;
    T H2, h, h_inv, Cgw, m0, Kddt, linfac, eta2, Kv2, rho0, inv_rho0, g, Cbe;
    T nhinv_half, Cfac, big;   // −1/(2h);  −8·Cgw: ∇W factor = Cfac·u³ with u = clamp(1 − q/2);  2⁴⁰ (step01)
    T Cbe7;                    // Cb/γ ÷ ρ₀⁷: Pressure! of a neighbour's ρ⁺ as ρ⁷·Cbe7 − Cb/γ (packed pair loop of the corrector; the host routes handles whose ρ₀⁷ leaves fp32 to the run-time variant)
    T inv_Kv2;                 // 1 / Kv2 (the compiled-in model accumulates in units of Kv2; handles whose Kv2 is not a normal number run the run-time variant)
    T alphaD, tens_eps, inv_Wdx;   // CubicSpline: αD, CubicSpline.eps, 1 / W(q := dx) (src/SPHKernels.jl:114-126)
    T Klam;              // 4·m₀·ν₀ (Laminar)
    T sps_cs2, sps_blin; // (Cs·dx)², (2/3)·C_Blin·dx² (LaminarSPS)
    double hyd_a, hyd_b; // ComplexDensityDiffusion: ρᴴ(z) = ρ₀·(⁷√(1 + hyd_a·z) − 1), hyd_a = ρ₀·g/Cb, hyd_b = ρ₀
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float  fast_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) {
    // v_rcp_f64 (≈2⁻²⁶ relative) + ONE Newton step: ≈2⁻⁵² for the normal, positive arguments of the pair loop (a second
    // step makes it ≤ 1 ulp and costs 4 % of the fp64 kernel; parity to the oracle is the same; IEEE division: 2.3e8 → 4.7e8 with this)
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ float  fast_sqrt(float x)  { return __builtin_amdgcn_sqrtf(x); }
// The two transcendentals of the fp32 pair (round 5), issued BACK TO BACK and followed by a scalar instruction: on gfx950 a transcendental among
// double-rate instructions (v_mul / v_add / v_fma: 2.6 cycles per wave64 instruction and SIMD) costs ≈12 cycles where it stands alone and 8.5 behind
// another one (tools/ubench/issue_patterns.hip, loop_replay.hip); the s_nop is also the wait state a VALU instruction needs before it reads a
// transcendental's result (the compiler cannot see into an asm).
__device__ __forceinline__ void sqrt_and_rcp(const float a, const float b, float& sqrt_a, float& rcp_b) {
#if defined(SPHMI_TRANS_SPLIT)
    asm("v_sqrt_f32 %0, %2\n\ts_nop 0\n\tv_rcp_f32 %1, %3\n\ts_nop 0" : "=&v"(sqrt_a), "=v"(rcp_b) : "v"(a), "v"(b));
#else
    asm("v_sqrt_f32 %0, %2\n\tv_rcp_f32 %1, %3\n\ts_nop 0" : "=&v"(sqrt_a), "=v"(rcp_b) : "v"(a), "v"(b));
#endif
}
#ifndef SPHMI_F64_TRIMS
#define SPHMI_F64_TRIMS 1       // fp64 pair loop (round 5): square root without the compiler's range scaling, the clamp of u as an output modifier, the neighbour's
                                // EOS with a multiply for its division — 0: rounds 1-4, A/B builds
#endif
__device__ __forceinline__ double fast_sqrt(double x) {
#if SPHMI_F64_TRIMS
    // sqrt of an r² of the pair loop (0 for the self pair, else 1e-8 … 1e-2 of a length² in metres: no subnormals, no infinities) — the compiler's sqrt() is
    // v_rsq_f64 + the same two refinements INSIDE a range scaling of two v_ldexp_f64, a v_cmp_class and four selects (17 instructions; this: 9).
    const double xs = x > 1e-300 ? x : 1e-300;
    const double y = __builtin_amdgcn_rsq(xs);
    double g = xs * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, xs);
    return __builtin_fma(d, h, g);
#else
    return sqrt(x);
#endif
}

__device__ __forceinline__ float  absT(float x)  { return __builtin_fabsf(x); }
__device__ __forceinline__ double absT(double x) { return __builtin_fabs(x); }

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double rl(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// min without the canonicalising v_max the compiler puts in front of fmin (inputs are never signalling NaNs)
__device__ __forceinline__ float min_raw(float a, float b) {
    float m; asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b)); return m;
}
__device__ __forceinline__ double min_raw(double a, double b) { return a < b ? a : b; }

// u = clamp(a·b + 1, 0, 1) in ONE instruction (output modifier): with a = r, b = −1/(2h) this is 1 − q/2 clamped, and
// (q − 2)³ of the Wendland gradient (src/SPHKernels.jl:85-86, q = clamp(r/h, 0, 2) of src/SPHCellList.jl:280) = −8u³
__device__ __forceinline__ float fma1_clamp01(float a, float b) {
    // s_nop: `a` is the result of v_sqrt_f32 one instruction earlier, and a VALU instruction that reads the result of a
    // transcendental needs one wait state on gfx950 — the compiler pads its own instructions but cannot see into an asm
    float r; asm("s_nop 0\n\tv_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(r) : "v"(a), "s"(b)); return r;
}
// (the same when `a` comes out of sqrt_and_rcp, whose trailing s_nop already is the wait state)
__device__ __forceinline__ float fma1_clamp01_ready(float a, float b) {
    float r; asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(r) : "v"(a), "s"(b)); return r;
}
__device__ __forceinline__ double fma1_clamp01(double a, double b) {
#if SPHMI_F64_TRIMS
    double r; asm("v_fma_f64 %0, %1, %2, 1.0 clamp" : "=v"(r) : "v"(a), "v"(b)); return r;      // (two compares and three selects as ONE output modifier)
#else
    const double t = __builtin_fma(a, b, 1.0);
    return t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
#endif
}
// 1 for a positive (Fluid) signed density, 0 for a negative one — the MotionLimiter of the neighbour as a factor
__device__ __forceinline__ float step01(float s, float big) {
    float r; asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(s), "s"(big)); return r;
}
__device__ __forceinline__ double step01(double s, double) { return s > 0.0 ? 1.0 : 0.0; }

// 16-/32-byte packet gathers through buffer loads: 32-bit offsets, one address instruction per gather
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// (jr = the neighbour's index × the record size, 32 or 64 bytes: the queue entries hold the candidate base pre-multiplied, so
// that base + bit·size is ONE v_lshl_add_u32; `half` = 0 / 1 selects the packet of the record)
__device__ __forceinline__ float4 gather_packet(__amdgpu_buffer_rsrc_t r, unsigned jr, int half, float) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(jr + 16u * half), 0, 0);
    float4 f; f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
    return f;
}
__device__ __forceinline__ double4 gather_packet(__amdgpu_buffer_rsrc_t r, unsigned jr, int half, double) {
    const u32x4_t lo = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(jr + 32u * half), 0, 0);
    const u32x4_t hi = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(jr + 32u * half + 16u), 0, 0);
    double4 f;
    f.x = __longlong_as_double(((long long)lo.y << 32) | lo.x); f.y = __longlong_as_double(((long long)lo.w << 32) | lo.z);
    f.z = __longlong_as_double(((long long)hi.y << 32) | hi.x); f.w = __longlong_as_double(((long long)hi.w << 32) | hi.z);
    return f;
}

// Estimate7thRoot, src/SimulationEquations.jl:49-61: Float64 bit trick + two Newton steps of t³ − x/t⁴ = 0
__device__ __forceinline__ double root7_estimate(double x) {
    const unsigned long long u = 0x36cd000000000000ull + (unsigned long long)__double_as_longlong(fabs(x)) / 7ull;
    double t = copysign(__longlong_as_double((long long)u), x);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, xot4 = x / t4;
        t = t - t * (t3 - xot4) / (4.0 * t3 + 3.0 * xot4);
    }
    return t;
}

// EquationOfStateGamma7 — src/SimulationEquations.jl:9-11
template <class T> __device__ __forceinline__ T eos7(T rho, T rho0, T inv_rho0, T Cbe) {
    T r;
    if constexpr (sizeof(T) == 8) r = rho / rho0; else r = rho * inv_rho0;
    T r2 = r * r, r4 = r2 * r2;
    return Cbe * (r4 * r2 * r - T(1));
}

template <class T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;                 // NaN in v survives (comparison false keeps v) —
        if (u != u) v = u;                 // — and NaN in u is taken explicitly
    }
    return v;
}
// max-reduction of non-negative values through their bit patterns (monotone for v ≥ 0; NaN sorts last).
// Almost every wave is below the running maximum already: test before paying for the atomic.
// (`pretest` = false, round 5: the launches of EIGHT waves per tile — below 330 tiles, a few hundred waves — send the atomic at once (with four waves per tile, 500-850 tiles, the atomics of 3 000 waves on three counters cost MovingSquare2d 9 % and DucklingMDBC 15 %): the pre-test is a
// device-scope load, ≈1 µs beyond the XCD's L2, that every wave WAITS for at the end of its life, and a launch of 108 tiles has nothing to
// save on 650 atomics; tools/trace_small.py: the corrector's epilogue 3.2 → … µs on the 2-D dam break)
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, float v, bool pretest = true) {
    unsigned int* q = reinterpret_cast<unsigned int*>(p);
    const unsigned int b = __float_as_uint(v);
    if (!pretest) { if (b != 0u) atomicMax(q, b); return; }
    if (b > __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(q, b);
}
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, double v, bool pretest = true) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (!pretest) { if (b != 0ull) atomicMax(p, b); return; }
    if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
}

// ------------------------------------------------------------------------------------------
// The neighbour + force kernel.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
// two floats -> one register of two halves, rounded towards zero (any rounding will do for a hi + lo split: lo takes what hi left)
__device__ __forceinline__ unsigned pk_h2(float a, float b) { const h16x2 p = __builtin_amdgcn_cvt_pkrtz(a, b); return __builtin_bit_cast(unsigned, p); }
__device__ __forceinline__ float lo_half_as_float(unsigned p) { return (float)__builtin_bit_cast(h16x2, p)[0]; }
// v = hi + lo + O(2^-20 |v|): { hi, lo } in one register ...
__device__ __forceinline__ unsigned split_hl(float v) { const unsigned h = pk_h2(v, 0.0f); return pk_h2(v, v - lo_half_as_float(h)); }
// ... and as { hi, hi }, { lo, lo }
__device__ __forceinline__ void split_hh_ll(float v, unsigned& hh, unsigned& ll) { hh = pk_h2(v, v); const float r = v - lo_half_as_float(hh); ll = pk_h2(r, r); }

// v_permlane32_swap: returns { {p.lower, q.lower}, {p.upper, q.upper} } — the lane pattern both MFMA
// operands want ("lanes 0-31: component k of item l, lanes 32-63: component k+1 of item l-32").
__device__ __forceinline__ void swap_halves(float& p, float& q) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(q), false, false);
    p = __uint_as_float(r[0]); q = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_halves(unsigned& p, unsigned& q) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(p, q, false, false);
    p = r[0]; q = r[1];
}

// LDS hand-off inside ONE wave (ds operations of a wave execute in order; this only pins the compiler)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MODEL >= 0: viscosity tag in bits 0-3, density-diffusion tag in bits 4-7, no shifting — compiled in (no
// uniform branches per pair; kModelDefault = ArtificialViscosity + LinearDensityDiffusion, what every stock
// example but one uses).  MODEL < 0: the tags are read from the parameter block at run time (all other
// combinations: Laminar / LaminarSPS, ZeroGravityLinear / Complex diffusion, PlanarShifting).
// WPT: waves per tile (1, 2, 4, 8).  The waves of a workgroup share the tile's 64 targets and split its candidate chunks
// round-robin over ALL rows (chunk c of a row whose predecessors hold g chunks goes to wave (g + c) % WPT: the waves'
// chunk counts differ by one at most); wave 0 adds the partial sums in wave order and runs the epilogue.  A wave's lifetime is the scheduling granule of a launch: with few tiles (small cases, and the
// last round of a 1 M-particle launch) shorter-lived waves keep the SIMDs filled.
// TPB: tiles per block.  The TPB tiles of a workgroup are TPB consecutive entries of the XCD's run —
// neighbouring tiles, whose candidate rows overlap by three quarters — and run on the four SIMDs of ONE compute unit,
// so the rows are fetched into that unit's L1 once instead of by four units.
template <class T, int D, int PASS, int MODEL, int WPT, int TPB = 1>
__global__ void __launch_bounds__(kWave * WPT * TPB)
__attribute__((amdgpu_waves_per_eu(1, 8)))
k_neighbor_force(const ForceParams<T> P) {
    static_assert(TPB == 1 || WPT <= 2, "several tiles per block: one or two waves per tile");
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    T step_dt, step_dt2;
    if (PASS == PASS_PREDICTOR && P.ctl_in != nullptr) {
        StepCtrl c = *P.ctl_in;
        const unsigned long long r0 = P.red_in[0], r1 = P.red_in[1], r2 = P.red_in[2], r3 = P.red_in[3];
        const bool consumed = step_control_decide<T>(r0, r1, r2, r3, c, P.ctl_h, P.ctl_c0, P.ctl_CFL);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *P.ctl_out = c;
            if (consumed) { P.red_zero[0] = 0; P.red_zero[1] = 0; P.red_zero[2] = 0; P.red_zero[3] = 0; }
        }
        if (!c.active) return;
        step_dt = (T)c.dt; step_dt2 = (T)c.dt2;
    } else {
        if (P.ctrl && !P.ctrl->active) return;             // a queued step that the control cancelled
        if (PASS == PASS_PREDICTOR && P.mdbc_zero != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
            P.mdbc_zero[0] = 0; P.mdbc_zero[1] = 0; P.mdbc_zero[2] = 0; *P.mdbc_flag_zero = 0;
        }
        step_dt = P.ctrl ? (T)P.ctrl->dt : P.dt; step_dt2 = P.ctrl ? (T)P.ctrl->dt2 : P.dt2;
    }
    const int visc = MODEL >= 0 ? (MODEL & 15) : P.visc;
    const int ddt = MODEL >= 0 ? ((MODEL >> 4) & 15) : P.ddt;
    const bool shift = MODEL >= 0 ? false : (P.shift != 0 && PASS == PASS_CORRECTOR);
    using V4 = typename Vec4<T>::type;
    constexpr int NSEG = (D == 3) ? 9 : 3;
    constexpr int QCAP = queue_entries<T, WPT, MODEL>();         // per-lane queue of non-empty accept masks
    constexpr int kQueueSlack = queue_slack<T, WPT, MODEL>();
    static_assert(QCAP >= 4 && kQueueSlack >= 1 && kQueueSlack <= QCAP - 1, "queue geometry");
    // entry = { 32-bit accept mask, record size × candidate index of its bit 0 }: 8 bytes, one ds_read_b64 per refill
    __shared__ uint2 s_q_all[SPHMI_LDS_STAGE ? 1 : WPT * TPB * QCAP * kWave];    // [wave][entry][lane]
    // SPHMI_LDS_STAGE: the two packets of the 64 candidates of the chunk being worked on, per wave (2 / 4 KB in fp32 / fp64)
    __shared__ V4 s_stage_all[SPHMI_LDS_STAGE ? WPT * TPB * 2 * kWave : 1];

    // SPHMI_DIAG == 16: the record ring whose price is being taken (fp32 half tiles of one wave per half, compiled-in models: the bench's kernels)
    constexpr bool kRing = SPHMI_DIAG == 16 && sizeof(T) == 4 && WPT == 2 && MODEL >= 0 && SPHMI_HALF_TILE != 0 && SPHMI_LDS_STAGE == 0;
    constexpr int kRingRec = SPHMI_RING_RECORDS;
    static_assert((kRingRec & (kRingRec - 1)) == 0 && kRingRec >= 64, "ring: a power of two of at least one chunk");
    __shared__ V4 s_ring_all[kRing ? 2 * kRingRec * (SPHMI_RING_SHARED ? 1 : WPT * TPB) : 1];

    const int lane = threadIdx.x & (kWave - 1);
    const int wvb = threadIdx.x >> 6;                      // wave of the block
    [[maybe_unused]] char* const s_ringb = reinterpret_cast<char*>(s_ring_all + (kRing && !SPHMI_RING_SHARED ? wvb * 2 * kRingRec : 0));
    // byte offsets of the two packets of the record whose offset in the record array is jr (= 32 × index)
    [[maybe_unused]] auto ring_off0 = [](const unsigned jr) -> unsigned { return SPHMI_RING_SOA ? ((jr >> 1) & (unsigned)(kRingRec * 16 - 1)) : (jr & (unsigned)(kRingRec * 32 - 1)); };
    [[maybe_unused]] constexpr unsigned kRingOff1 = SPHMI_RING_SOA ? (unsigned)kRingRec * 16u : 16u;
    // wave of the tile, tile of the block.  Two tiles of two waves (the launches that fit the chip at once): a workgroup of
    // FOUR waves puts one wave on each SIMD of its compute unit, so every SIMD of a unit holds the same number of waves —
    // workgroups of two left the SIMDs with 4 … 9 waves and 35 % more work on the fullest than on average, and the launch
    // lasts as long as its fullest SIMD (tools/trace_waves.py, 158 791 particles).  Odd blocks swap the roles of a tile's
    // two waves, so that no SIMD always gets the wave that also runs the epilogue.
    const int tib = TPB == 1 ? 0 : wvb / WPT;
    const int wv = TPB == 1 ? wvb : (WPT == 1 ? 0 : ((wvb % WPT) ^ (int)((blockIdx.x >> 3) & 1)));
    // every lane owns one column of the queue array: no lane ever reads another lane's entries, so
    // program order is all the synchronisation the queue needs
    uint2* const s_q = s_q_all + (SPHMI_LDS_STAGE ? 0 : wvb * QCAP * kWave + lane);
    [[maybe_unused]] V4* const s_stage = s_stage_all + (SPHMI_LDS_STAGE ? wvb * 2 * kWave : 0);
    // Tile schedule (sphmi_rebuild.h): the dispatcher places block b on XCD b % 8; every XCD works through
    // one contiguous, cost-balanced run of tiles, expensive tiles first.  Measured on the 1 M-particle dam
    // break: equal-count contiguous runs 1.02 ms, 64-tile round-robin chunks 1.10 ms, identity 1.13 ms.
    constexpr bool kHalf = (WPT == 2 || ((WPT == 4 || WPT == 8) && SPHMI_HALF4 != 0)) && SPHMI_HALF_TILE != 0 && SPHMI_LDS_STAGE == 0;
    constexpr int kPar = kHalf ? WPT / 2 : 1;               // waves per half tile
    // waves of one workgroup may leave early for DIFFERENT reasons while others go on to a workgroup barrier (see below)
    constexpr bool kMixedExit = (kHalf && kPar > 1) || (!kHalf && WPT > 1 && TPB > 1);
    bool dead = false;
    int b;
    {
        const int x = blockIdx.x & 7, r = TPB == 1 ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 3) * TPB + tib;
        // (Early exits and the workgroup barriers further down.  A cancelled step ends every wave of the launch; an exhausted XCD run or a
        // tile of ghosts only ends the waves of ONE tile — or, with half tiles, of one HALF.  Where such waves share a workgroup with waves
        // that go on to a `__syncthreads()` (`kMixedExit`: half tiles worked off by two or four waves, and the chunk-splitting builds with
        // two tiles per block) they do not return: they are marked `dead`, scan no row, queue no pair, store nothing, and reach every
        // barrier of the block like the others — a barrier that not every thread of the block reaches is undefined in the HIP model,
        // whatever gfx9's `s_endpgm` does to the barrier count (rounds 2-4 relied on that).  Everywhere else the exit is uniform over
        // the workgroup or the kernel has no barrier, and the waves simply return.
        // test_every_waves_per_tile_variant_matches_the_oracle and tests/test_multi_gpu.py force every WPT / TPB variant on layouts
        // with partial last blocks and ghost-only tiles.)
        if (r >= P.part[8 + x]) {
            if constexpr (TPB == 1 || !kMixedExit) return;      // (one tile per block: the whole workgroup leaves here)
            dead = true;
        }
        b = dead ? 0 : P.order[P.part[x] + r];
    }
    // kHalf (tiles of two waves): wave w serves targets 32w … 32w+31 of the tile, TWO LANES PER TARGET — lane l and lane l + 32 hold the
    // same target and take the lower / upper 32 candidates of every chunk (what the matrix layout hands each lane half anyway: one
    // target block per chunk instead of two, no exchange of mask halves), and add their sums once after the loop.  The two waves of a
    // tile share nothing — no partial sums in LDS, no barrier — and a target's pairs are dealt half-chunk by half-chunk instead of chunk
    // by chunk: 159 k particles ran 93.8 pair-loop iterations per wave for 67.6 pairs per lane (72 % of the lane slots; one wave per
    // tile: 90 %).
    // Four-wave tiles (launches of 512 … 850 tiles) are two half tiles of TWO waves: the waves of a half deal its chunks alternately and the
    // second hands its sums to the first through LDS — the lanes of a target still share every chunk they scan.
    const int hl = kHalf ? (lane >> 5) : 0;                 // which 32 candidates of a chunk this lane takes
    const int half = kHalf ? (wv & 1) : 0;                  // which half of the tile this wave serves
    [[maybe_unused]] const int par = kHalf ? (wv >> 1) : 0; // … and which of the half's waves it is
    const int t0 = kHalf ? b * kWave + 32 * half : b * kWave;
    const int a = kHalf ? t0 + (lane & 31) : t0 + lane;
    const bool valid = a < P.N;
    const int ac = valid ? a : P.N - 1;
    // ghost copies (type bits 0xC0: owned by a neighbour rank) take part as neighbours only: their own
    // state arrives by halo exchange, so they get no accept masks, and nothing is stored or reduced for them
    const uint8_t ty_raw = P.type[ac];
    const bool owned = !dead && valid && !(ty_raw & 0xC0);
    unsigned long long owned_lanes = __builtin_amdgcn_ballot_w64(owned);
    if (owned_lanes == 0) {                                       // a tile (a half) of ghosts only
        if constexpr (!kMixedExit) return;
        dead = true; owned_lanes = 1;                             // (lane_o below: any lane — a dead wave computes nothing that is kept)
    }

    // target data
    const V4 q0 = P.src0[ac];
    const V4 q1 = P.src1[ac];
    const T xa = q0.x, ya = q0.y, za = q0.z;
    if constexpr (kRing) {
        // finite values in every slot (the lane's own record): what the loop reads is wrong, but it is not a NaN from a previous kernel's LDS
        for (int k = lane; k < kRingRec; k += kWave) {
            *reinterpret_cast<V4*>(s_ringb + ring_off0((unsigned)k << 5)) = q0;
            *reinterpret_cast<V4*>(s_ringb + ring_off0((unsigned)k << 5) + kRingOff1) = q1;
        }
        // (a shared ring: every wave fills ALL of it — waves of a block may have left already, so no barrier; whoever writes, the values are finite)
    }
    T rho_a, rhon_a, P_a, s_a;
    if constexpr (PASS == PASS_CORRECTOR) {
        rho_a = q0.w;                                   // ρ⁺
        rhon_a = absT(q1.w);                            // SimParticles.Density (quirk Q2)
        s_a = q1.w;
        P_a = eos7<T>(rho_a, P.rho0, P.inv_rho0, P.Cbe);
    } else {
        rho_a = absT(q0.w);
        rhon_a = rho_a;
        s_a = q0.w;
        P_a = q1.w;                                     // Pressure! ran before mDBC (quirk Q3)
    }
    // (round 5, launches of four and eight waves per tile: the corrector's epilogue reads state A and the low words of this lane's particle — requested
    // HERE, a whole pass ahead of their use, instead of as one more exposed round trip at the end of a wave whose life is 10 µs; twelve
    // registers that those kernels have to spare, the large launches do not)
    // (compiled-in models only: the run-time-model corrector of eight waves per tile sits at 127 registers, and twelve more cost it its second workgroup per
    // compute unit — +5 … +14 % per step on the 17 k-particle Laminar / SPS / shifting handles, caught by tools/variants_vs_previous.sh)
    constexpr bool kEarlyEpilogueLoads = PASS == PASS_CORRECTOR && WPT >= 4 && MODEL >= 0 && SPHMI_SMALL_TRIMS != 0;
    [[maybe_unused]] V4 pre_s0, pre_s1, pre_lo;
    if constexpr (kEarlyEpilogueLoads) {
        pre_s0 = P.a0[ac]; pre_s1 = P.a1[ac];
        pre_lo.x = pre_lo.y = pre_lo.z = pre_lo.w = T(0);
        if (sizeof(T) == 4 && P.comp != nullptr) pre_lo = P.comp[ac];
    }
    const bool fluid_a = s_a > T(0);
    const T inv_rho_a = fast_rcp(rho_a);
    const T inv_rhon_a = (PASS == PASS_CORRECTOR) ? fast_rcp(rhon_a) : inv_rho_a;
    const T rm_a = rho_a * P.m0;
    // lane constants of the pair terms: −m₀/ρₐ (pressure), −2·δᵩhc₀m₀·MLₐ (density diffusion; ZeroGravityLinear has no
    // MotionLimiter factor), Pₐ − Cb/γ·… (corrector: Pₐ + P_b = Cbe·r_b⁷ + (Pₐ − Cbe))
    // (the compiled-in model accumulates a / (Kv2·Cfac) and the density sums / Cfac — the viscosity term loses its
    // constant, the pressure term takes 1/Kv2 into its lane constant, ∇W's factor is the bare u³; the sums are scaled back once
    // after the loop; the host routes α = 0 to the run-time variant)
    constexpr bool kFoldKv2 = MODEL >= 0 && (MODEL & 15) == kViscArtificial && sizeof(T) == 4;
    const T c_a = kFoldKv2 ? -P.m0 * inv_rho_a * P.inv_Kv2 : -P.m0 * inv_rho_a;
    const T Kd_a = (ddt == kDdtZeroGravityLinear || fluid_a) ? T(-2) * P.Kddt : T(0);
    const T PaC = P_a - P.Cbe;

    const int key_a = P.key[ac];
    const int cs_a = P.cstart[key_a], ce_a = P.cstart[key_a + 1];
    const int last_lane = min(kHalf ? 31 : kWave - 1, P.N - 1 - t0);

    // Phase 1 works in tile-local coordinates (origin = the tile's first particle) and in the expanded
    // form  |c − t|² − H'² = |c|² − 2c·t + (|t|² − H'²)  with a slightly generous cut-off
    // H'² = H²(1+ε); ε covers the fp32 cancellation error of the expanded form (≈ 4·2⁻²⁴·R², R = largest
    // local coordinate).  Phase 2 redoes the exact r² ≤ H² test of src/SPHCellList.jl:275.
    // The mask only has to be a SUPERSET of the pairs within H (the pair loop is exact), so fp64 handles run the same
    // fp32 matrix: local coordinates are formed in T and rounded to fp32 (2.27 → ≈1.8 ms per launch at 1 M particles
    // against one target per iteration through SGPRs on the fp64 vector ALU).
    // (The origin is the first OWNED particle of the tile, and a ghost copy takes part with local coordinates 0: the interior launch of
    // a slab runs BEFORE the halo of the pass has landed, a tile may hold ghost rows next to owned ones, and what those rows contain then
    // is last step's state — or, in a record set that has never been written, whatever the allocation held: a NaN there made the origin,
    // the reach and with it the cut-off of every lane NaN, and the whole tile lost or gained all its candidates by the sign of that NaN.
    // Found by a fresh generation of tests/test_fuzz_gpu.py: three slabs, first steps of a new handle, intermittent.)
    const int lane_o = __builtin_ctzll(owned_lanes);
    const T ox = rl(xa, lane_o), oy = rl(ya, lane_o), oz = rl(za, lane_o);
    const float txl = owned ? (float)(xa - ox) : 0.0f, tyl = owned ? (float)(ya - oy) : 0.0f, tzl = owned ? (float)(za - oz) : 0.0f;
    const float tt = txl * txl + tyl * tyl + tzl * tzl;
    float thr;
    {
        const float H2f = (float)P.H2;
        const float Rm = fast_sqrt(wave_max(owned ? tt : 0.0f)) + 6.0f * (float)P.h;
        const float eps = 1e-5f + 1e-6f * (Rm * Rm) / H2f;
        thr = owned ? H2f * (1.0f + eps) - tt : -1e30f;
    }
    [[maybe_unused]] const float m2x = -2.0f * txl, m2y = -2.0f * tyl, m2z = -2.0f * tzl;

    T drho = 0, ax = 0, ay = 0, az = 0;
    T gcx = 0, gcy = 0, gcz = 0, divr = 0;            // PlanarShifting: ∇Cᵢ, ∇◌rᵢ (corrector pass)
    T kgx = 0, kgy = 0, kgz = 0, kw = 0;              // StoreKernelOutput: Σ∇W, ΣW (corrector pass)
    V4 vn_a = q1;                                       // SimParticles.Velocity of the target (LaminarSPS)
    if constexpr (PASS == PASS_CORRECTOR) { if (visc == kViscLaminarSPS) vn_a = P.a1[ac]; }

    // SimParticles.Velocity of the neighbours (LaminarSPS in the corrector pass)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.a0.p, 0, (int)((unsigned)P.N * 2u * (unsigned)sizeof(V4)), 0x00020000);
    // ---- pair physics for one accepted neighbour j ------------------------------------------
    T sum_c = 0, sum_d = 0;                             // Σ (1/ρ_b)·(∇W·vᵢⱼ) (continuity without ρₐm₀), Σ density diffusion
    constexpr int kRecShift = sizeof(T) == 4 ? 5 : 6;      // log2 of the record size
    // bit p of a queue entry → the candidate's distance from the entry's base.  Half tiles with interleaved shares: the matrix layout
    // hands lane half h the rows 8g + 4h + k (g = 0 … 7, k = 0 … 3) of a chunk loaded in natural order, so bit p = 4g + k is candidate
    // 8g + k = p + (p & ~3) behind the base cb + 4h — the two lanes of a target take alternate groups of four candidates and their
    // pair counts differ by a handful instead of by half a cell (one v_and + one v_add per pair)
    constexpr bool kInterleave = kHalf && SPHMI_HALF_INTERLEAVE != 0;
    // record offset of the candidate at the lowest set bit of m, for an entry whose bit 0 sits at record offset `base`
    auto rec_of = [](const unsigned m, const unsigned base) -> unsigned {
        const unsigned p = (unsigned)__builtin_ctz(m);
        if constexpr (kInterleave) return ((p + (p & 0x1Cu)) << kRecShift) + base;
        else return (p << kRecShift) + base;
    };
    // (record offsets of the target's cell [cs_ar, ce_ar) and of the record BEHIND the target, a_r1: what the orientation rule and the kernel output compare with)
    const unsigned cs_ar = (unsigned)cs_a << kRecShift, a_r1 = ((unsigned)a + 1u) << kRecShift, w_ai = ((unsigned)ce_a << kRecShift) - a_r1;
    // `if_i` when the target plays "i", `if_j` otherwise
    auto pick_i = [&](const T if_i, const T if_j, const bool a_is_i) -> T { return a_is_i ? if_i : if_j; };
    auto pair_core = [&](const unsigned jr, const T dx, const T dy, const T dz, const T n0w, const V4& n1, const bool a_is_i) {
        const T r2 = (D == 3) ? dx * dx + dy * dy + dz * dz : dx * dx + dy * dy;
        T rho_b, rhon_b, s_b;
        if constexpr (PASS == PASS_CORRECTOR) { rho_b = n0w; rhon_b = absT(n1.w); s_b = n1.w; }
        else { rho_b = absT(n0w); rhon_b = rho_b; s_b = n0w; }
        // ∇W factor, src/SPHKernels.jl:80-87 with q = clamp(r/h, 0, 2) (src/SPHCellList.jl:280): (q − 2)³ = −8u³,
        // u = clamp(1 − q/2, 0, 1) — one fused multiply-add with the clamp output modifier.
        // The phase-1 mask is slightly generous; with H = 2h (the default k = 2) the r² ≤ H² cut of :275 needs no
        // branch: beyond H the clamp makes u = 0 and every pair term below carries the factor `fac`.
        const T r = fast_sqrt(r2);
        const T u = fma1_clamp01(r, P.nhinv_half);
        T fac = kFoldKv2 ? (u * u) * u : P.Cfac * (u * u * u);       // (folded: every sum of the compiled-in model is linear in Cfac — applied after the loop)
        T Wq = T(0);                                            // W(q): tensile correction / kernel output
        const bool cubic = MODEL < 0 && P.kernel == 1;
        if (MODEL < 0 && (cubic || P.kout)) {
            const T q = min_raw(r * P.h_inv, T(2)), tq = q - T(2);
            if (cubic) {
                // CubicSpline, src/SPHKernels.jl:89-106: ∇W = dW/dq·h⁻¹·xᵢⱼ/(|xᵢⱼ| + η²)
                const T dWdq = q <= T(1) ? P.alphaD * (T(-3) * q + T(2.25) * q * q) : P.alphaD * T(-0.75) * (tq * tq);
                fac = dWdq * P.h_inv / (r + P.eta2);
                Wq = q <= T(1) ? P.alphaD * (T(1) - T(1.5) * q * q + T(0.75) * q * q * q) : P.alphaD * T(0.25) * (-(tq * tq * tq));
            } else {
                const T t1 = T(1) - q * T(0.5), t2 = t1 * t1;
                Wq = P.alphaD * (t2 * t2) * (T(2) * q + T(1));
            }
        }
        // H = k·h with k < 2 (example/DucklingMDBC.jl: 1.5, MovingSquare2d.jl: √2) cuts the kernel off before it
        // vanishes: there the cut of :275 has to be applied for real (run-time variant of the kernel only)
        if ((MODEL < 0 && P.exact_cut) || (MODEL >= 0 && (MODEL & kModelCutBit) != 0)) fac = (r2 <= P.H2) ? fac : T(0);
        const T dvx = q1.x - n1.x, dvy = q1.y - n1.y, dvz = (D == 3) ? q1.z - n1.z : T(0);
        const T vdx = (D == 3) ? dvx * dx + dvy * dy + dvz * dz : dvx * dx + dvy * dy;          // vᵢⱼ·xᵢⱼ
        const T inv_rho_b = fast_rcp(rho_b);
        // continuity, src/SPHCellList.jl:289-291 (both orientations give the same target term); ρₐm₀ after the loop
        sum_c += inv_rho_b * (fac * vdx);
        constexpr bool kProdRcp = sizeof(T) == 4 && MODEL >= 0 && (MODEL & 15) == kViscArtificial;
        T inv_r2e, inv_r2e_rs = T(0);                          // 1/(r²+η²);  1/((r²+η²)(ρ̄ₐ+ρ̄_b)) (artificial viscosity)
        if constexpr (kProdRcp) {
            const T rs = rhon_a + rhon_b;
            inv_r2e_rs = fast_rcp((r2 + P.eta2) * rs);
            inv_r2e = inv_r2e_rs * rs;
        } else inv_r2e = fast_rcp(r2 + P.eta2);
        if (ddt != kDdtNone) {
            // density diffusion, src/SPHDensityDiffusionModels.jl:56-87 (no hydrostatic part, no MLcond),
            // :100-136 (linear), :150-188 (inverse hydrostatic EOS); orientation rule of SURVEY §8(a)-Q4:
            // the target plays "i" iff j sorts before its cell, or after it inside it (a_is_i, kept by the caller)
            const T dlast = (D == 3) ? dz : dy;
            T rhoH = T(0);
            if (ddt == kDdtLinear) rhoH = P.linfac * dlast;
            else if (ddt == kDdtComplex) {
                // as "i" the pair sees ρᴴ(xᵢⱼ[end]); as "j" it sees −ρᴴ(−xᵢⱼ[end]) of the mirrored pair
                const double z = a_is_i ? (double)dlast : -(double)dlast;
                const double rh = P.hyd_b * (root7_estimate(1.0 + P.hyd_a * z) - 1.0);
                rhoH = (T)(a_is_i ? rh : -rh);
            }
            const T drn = (rhon_b - rhon_a) - rhoH;
            T inv_sel;
            if constexpr (PASS == PASS_CORRECTOR) inv_sel = pick_i(fast_rcp(rhon_b), inv_rhon_a, a_is_i);
            else inv_sel = pick_i(inv_rho_b, inv_rho_a, a_is_i);
            // Dᵢ = δᵩhc₀·(m₀/ρ_sel)·ψ·∇W·MLᵢMLⱼ with ψ·∇W = −2·Δρ·fac·r²/(r²+η²); MLᵢ sits in Kd_a, MLⱼ is a 0/1 factor
            const T Dv = (Kd_a * inv_sel) * (drn * (fac * (r2 * inv_r2e)));
            const T on = ddt == kDdtZeroGravityLinear ? T(1) : step01(s_b, P.big);
            sum_d += Dv * on;
        }
        // pressure, src/SPHCellList.jl:301-303 (tensile term is 0 for Wendland): −m₀(Pᵢ+Pⱼ)/(ρᵢρⱼ)
        T Psum, P_b;
        if constexpr (PASS == PASS_CORRECTOR) {
            // EquationOfStateGamma7 (src/SimulationEquations.jl:9-11) of the neighbour's ρ⁺, folded into the sum
            T rr;
            // (fp64: ρ/ρ₀ as in src/SimulationEquations.jl:10 costs an IEEE division — a dozen instructions — per PAIR.  Division by a CONSTANT: the
            // quotient through the reciprocal, corrected with the exact residual — two FMAs, and the correctly rounded ρ/ρ₀ again but for rare double
            // roundings; a bare multiply is off by an ulp often enough to move MovingSquare2d's r = H ties, 4.6e-13 → 3e-8 against the oracle after 100 steps)
            if constexpr (sizeof(T) == 8 && SPHMI_F64_TRIMS == 0) rr = rho_b / P.rho0;
            else if constexpr (sizeof(T) == 8) { const T q0 = rho_b * P.inv_rho0; rr = __builtin_fma(__builtin_fma(-q0, P.rho0, rho_b), P.inv_rho0, q0); }
            else rr = rho_b * P.inv_rho0;
            const T rr2 = rr * rr, rr4 = rr2 * rr2, rr7 = (rr4 * rr2) * rr;
            Psum = P.Cbe * rr7 + PaC;
            P_b = Psum - P_a;
        } else { P_b = n1.w; Psum = P_a + P_b; }
        T coef = Psum * (c_a * inv_rho_b);
        if (cubic) {
            // tensile_correction, :114-126 (n = 4; the reference evaluates the reference kernel value at q := dx)
            const T w = Wq * P.inv_Wdx, w2 = w * w;
            coef -= P.m0 * (P.tens_eps * ((P_a * inv_rho_a * inv_rho_a) + (P_b * inv_rho_b * inv_rho_b)) * (w2 * w2));
        }
        if (visc == kViscArtificial) {
            // ArtificialViscosity, src/SPHViscosityModels.jl:56-74 (ρ̄ from SimParticles.Density)
            const T vneg = min_raw(vdx, T(0));
            if constexpr (kProdRcp) coef += kFoldKv2 ? vneg * inv_r2e_rs : P.Kv2 * (vneg * inv_r2e_rs);
            else coef += kFoldKv2 ? (vneg * inv_r2e) * fast_rcp(rhon_a + rhon_b) : (P.Kv2 * (vneg * inv_r2e)) * fast_rcp(rhon_a + rhon_b);
        }
        coef *= fac;
        ax += coef * dx; ay += coef * dy;
        if constexpr (D == 3) az += coef * dz;
        if (visc == kViscLaminar || visc == kViscLaminarSPS) {
            // Laminar, :77-87: term·vᵢⱼ with term = 4m₀ν₀(xᵢⱼ·∇W)/((ρᵢ+ρⱼ) + (d²+η²)) — the reference ADDS the
            // two brackets; ρ from SimParticles.Density
            const T lam = P.Klam * (fac * r2) / ((rhon_a + rhon_b) + (r2 + P.eta2));
            ax += lam * dvx; ay += lam * dvy; az += lam * dvz;
            if (visc == kViscLaminarSPS) {
                // LaminarSPS, :90-126, with SimParticles.Velocity / .Density of BOTH passes.  Both strain
                // tensors are multiples of O = (vⱼ−vᵢ)⊗∇W:  Sᵢ = (m₀/ρⱼ)O, Sⱼ = (m₀/ρᵢ)O.
                T wx, wy, wz;                                   // vⁿ_b − vⁿ_a
                if constexpr (PASS == PASS_CORRECTOR) {
                    const V4 nv = gather_packet(rsA, jr, 1, T());
                    wx = nv.x - vn_a.x; wy = nv.y - vn_a.y; wz = nv.z - vn_a.z;
                } else { wx = -dvx; wy = -dvy; wz = -dvz; }
                const T gx = fac * dx, gy = fac * dy, gz = fac * dz;
                const T gg = gx * gx + gy * gy + gz * gz, ww = wx * wx + wy * wy + wz * wz;
                const T trO = wx * gx + wy * gy + wz * gz;
                const T normO = fast_sqrt(T(2) * ww * gg);
                const T ka = P.m0 * fast_rcp(rhon_b), kb = P.m0 * inv_rhon_a;   // Sᵢ = ka·O, Sⱼ = kb·O
                // τ = 2·νt·ρ·(S − tr(S)/3·I) − (2/3)·ρ·C_B·dx²·‖S‖²·I,  νt = (Cs·dx)²·‖S‖,  ‖S‖ = k·normO
                const T na = ka * normO, nb = kb * normO;
                const T ca = T(2) * (P.sps_cs2 * na) * rhon_a, cb = T(2) * (P.sps_cs2 * nb) * rhon_b;
                // (τᵢ + τⱼ)·∇W = [ca·ka + cb·kb]·(gg·w − trO/3·g) − sps_blin·(ρᵢ·na² + ρⱼ·nb²)·g
                const T c1 = ca * ka + cb * kb;
                const T c2 = c1 * (trO * (T(1) / T(3))) + P.sps_blin * (rhon_a * (na * na) + rhon_b * (nb * nb));
                const T pre = P.m0 * inv_rhon_a * fast_rcp(rhon_b);
                ax += pre * (c1 * gg * wx - c2 * gx);
                ay += pre * (c1 * gg * wy - c2 * gy);
                az += pre * (c1 * gg * wz - c2 * gz);
            }
        }
        if (MODEL < 0 && P.kout && PASS == PASS_CORRECTOR) {
            // KernelOutput!, src/SPHCellList.jl:106-116
            const bool in = (r2 <= P.H2) && (jr + (1u << kRecShift) != a_r1);           // the pair loop never meets i == j
            kw += in ? Wq : T(0);
            kgx += fac * dx; kgy += fac * dy; kgz += fac * dz;
        }
        if (shift) {
            // add_shifting_terms!, src/SPHCellList.jl:73-88 (loop densities; both orientations give these)
            const T k = P.m0 * inv_rho_a * fac;
            gcx += k * dx; gcy += k * dy; gcz += k * dz;
            const T dv = P.m0 * inv_rho_b * (-(fac * r2));
            divr += (fluid_a && s_b > T(0)) ? dv : T(0);
        }
    };
    // ---- the same pair for the fp32 kernels of the compiled-in models (ArtificialViscosity + LinearDensityDiffusion, kFoldKv2): round 5.
    // Same terms as pair_core; what differs is how they are evaluated (profiles/HISTORY.md §4.9):
    //  * ONE reciprocal per pair: inv = 1/(ρ_b·(r²+η²)(ρ̄ₐ+ρ̄_b)) [corrector: ·ρⁿ_b as well]; 1/ρ_b, 1/((r²+η²)ρ̄) [and 1/ρⁿ_b] are inv times the
    //    other factors — two (three) transcendentals become multiplies, and the remaining two are issued back to back (sqrt_and_rcp);
    //  * the density-diffusion sum runs without its lane constant Kd_a (applied once after the loop);
    //  * Pressure! of a neighbour's ρ⁺ (corrector) as ρ⁷·(Cb/γ/ρ₀⁷) − Cb/γ: no scaling multiply;
    //  (packed fp32 for the head of the pair and the accumulators — 47 instead of 56 vector instructions — was built and measured 1.3 % slower: on gfx950 v_fma_f32 runs at
    //  the double rate and v_pk_* at the full rate; profiles/r05_retired_switches.patch, profiles/HISTORY.md §4.9)
    constexpr bool kFast = kFoldKv2 && MODEL >= 0 && ((MODEL >> 4) & 15) == kDdtLinear && SPHMI_FAST_PAIR != 0;
    constexpr bool kFastDiag = SPHMI_DIAG == 0 || SPHMI_DIAG == 2 || SPHMI_DIAG == 5 || SPHMI_DIAG == 8 || SPHMI_DIAG == 16;      // (the diagnostic builds that keep the arithmetic)
    [[maybe_unused]] auto pair_fast = [&](const V4& n0, const V4& n1, const bool a_is_i) {
        if constexpr (kFast) {
            float dz = 0.0f, r2, vdx;
            const float dx = xa - n0.x, dy = ya - n0.y;
            const float dvx = q1.x - n1.x, dvy = q1.y - n1.y;
            if constexpr (D == 3) {
                dz = za - n0.z;
                const float dvz = q1.z - n1.z;
                r2 = dx * dx + dy * dy + dz * dz; vdx = dvx * dx + dvy * dy + dvz * dz;
            } else { r2 = dx * dx + dy * dy; vdx = dvx * dx + dvy * dy; }
            float rho_b, rhon_b, s_b;
            if constexpr (PASS == PASS_CORRECTOR) { rho_b = n0.w; rhon_b = absT(n1.w); s_b = n1.w; }
            else { rho_b = absT(n0.w); rhon_b = rho_b; s_b = n0.w; }
            const float rs = rhon_a + rhon_b;
            const float prod = (r2 + P.eta2) * rs;
            float r, inv, inv_rho_b, inv_rhon_b, inv_r2e_rs;
            if constexpr (PASS == PASS_CORRECTOR) {
                const float qq = rho_b * rhon_b;
                sqrt_and_rcp(r2, qq * prod, r, inv);
                const float pq = inv * prod;
                inv_rho_b = pq * rhon_b; inv_rhon_b = pq * rho_b; inv_r2e_rs = inv * qq;
            } else {
                sqrt_and_rcp(r2, rho_b * prod, r, inv);
                inv_rho_b = inv * prod; inv_rhon_b = inv_rho_b; inv_r2e_rs = inv * rho_b;
            }
            const float u = fma1_clamp01_ready(r, P.nhinv_half);
            float fac = (u * u) * u;
            if constexpr ((MODEL & kModelCutBit) != 0) fac = (r2 <= P.H2) ? fac : 0.0f;
            const float inv_r2e = inv_r2e_rs * rs;
            const float cairb = c_a * inv_rho_b;
            // density diffusion, src/SPHDensityDiffusionModels.jl:100-136 with the orientation rule Q4 (Kd_a after the loop)
            const float drn = (rhon_b - rhon_a) - P.linfac * (D == 3 ? dz : dy);
            const float inv_sel = a_is_i ? inv_rhon_b : inv_rhon_a;
            const float Dv = inv_sel * (drn * (fac * (r2 * inv_r2e)));
            const float on = step01(s_b, P.big);
            sum_c += inv_rho_b * (fac * vdx); sum_d += Dv * on;
            // pressure (src/SPHCellList.jl:301-303) + ArtificialViscosity (src/SPHViscosityModels.jl:56-74)
            float Psum;
            if constexpr (PASS == PASS_CORRECTOR) {
                const float x2 = rho_b * rho_b, x3 = x2 * rho_b, x6 = x3 * x3, x7 = x6 * rho_b;
                Psum = x7 * P.Cbe7 + PaC;
            } else Psum = P_a + n1.w;
            const float vneg = min_raw(vdx, 0.0f);
            float coef = Psum * cairb + vneg * inv_r2e_rs;
            coef *= fac;
            ax += coef * dx; ay += coef * dy;
            if constexpr (D == 3) az += coef * dz;
        }
    };
    // 2-D handles keep z = vz = 0: the z terms are dropped at compile time
    auto pair = [&](const unsigned jr, const V4& n0, const V4& n1, const bool a_is_i) {
        if constexpr (kFast && kFastDiag) { pair_fast(n0, n1, a_is_i); return; }
#if SPHMI_DIAG == 1 || SPHMI_DIAG == 4
        // DIAGNOSTIC BUILD (wrong results): the gathers without the arithmetic — the floor set by the gather path
        sum_c += n0.x + n1.x;
#else
        pair_core(jr, xa - n0.x, ya - n0.y, (D == 3) ? za - n0.z : T(0), n0.w, n1, a_is_i);
#endif
    };

    // ---- phase 2: every lane walks the set bits of its own accept masks -----------------------
    // ONE descriptor over the neighbour records (2 packets each): 32-bit byte offsets, N·2·sizeof(packet) < 4 GB
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)P.src0.p, 0, (int)((unsigned)P.N * 2u * (unsigned)sizeof(V4)), 0x00020000);
    const unsigned long long xcd_t0 = P.xcd_clock ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : 0ull;
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SPHMI_STATS
    unsigned long long st_it = 0, st_lane = 0, st_ref = 0, st_emp = 0, st_chunks = 0;
#endif
    int work_it = 0, work_ch = 0;    // wave-uniform work counters (scalar unit): pair-loop iterations, chunks scanned
    // this lane's queue: a ring of QCAP entries in the lane's LDS column (entry k sits k·64·8 bytes further).  waddr / raddr
    // are the BYTE offsets of the next entry to write / fetch, wrapped with one subtract + unsigned min (QCAP need not be a
    // power of two: LDS per wave decides it); qn counts the queued entries.
    constexpr unsigned kEntryStride = kWave * 8, kQBytes = QCAP * kEntryStride;
    auto q_next = [](unsigned a) -> unsigned { a += kEntryStride; const unsigned b = a - kQBytes; return b < a ? b : a; };
    unsigned waddr = 0, raddr = 0;
    int qn = 0;
    unsigned cbase = 0;              // record offset of the candidate at bit 0 of the current mask
    unsigned cm = 0;                 // unconsumed bits of the current mask
    char* const s_qb = reinterpret_cast<char*>(s_q);
    // Phase 2 runs until no lane holds more than `keep` queued entries (`drain`: nor any fetched bit).
    // Lanes consume at their own pace: a lane fetches its next NON-EMPTY mask the moment its current one
    // is used up, so nobody waits for a neighbour lane and nobody spends an iteration on an empty mask.
    // Lone waves (a tile of four or eight waves = a launch too small to hide latency behind other waves) take TWO neighbours per
    // iteration (measured, µs per step one → two pairs: 2-D dam break 35.1 → 32.5, Dambreak3d Dp0.02 84.8 → 76.6, MovingSquare2d
    // 53.0 → 49.4; the 3-D run-time-model kernel at four waves per tile — DucklingMDBC — loses, 147.6 → 157.8: its corrector has
    // no registers left for a second neighbour)
    // (re-measured with half tiles of two / four waves per half: one pair per iteration is 0 … 7 % slower from 108 to 2 482 tiles, fp64 most)
    constexpr bool kTwoPairs = WPT >= 8 || (WPT >= 4 && (MODEL >= 0 || D == 2));
    // orientation of the density-diffusion term: the target plays "i" iff j sorts before its
    // cell (j < cs_a) or after it inside it (a < j < ce_a)
    // (the range a < j < ce_a as ONE unsigned compare of j − (a + 1) against ce_a − (a + 1): a subtract and two compares instead of three compares)
    auto plays_i = [&](const unsigned jr) { return (bool)((jr < cs_ar) | ((jr - a_r1) < w_ai)); };
    // "current mask used up AND something queued" is ONE unsigned compare, cm < qf with qf = min(qn, 1): the 0 / 1 flag is kept up
    // to date where qn changes (a refill, the end of a chunk's pushes) instead of two compares per iteration.  The loop tests are
    // computed once per iteration, at its END (a hand-rotated do … while: written top-tested the compiler copied six accumulators
    // per iteration).
    auto run_pairs_plain = [&](const int keep, const bool drain, auto two_tag) __attribute__((always_inline)) {
        constexpr bool kTwo = decltype(two_tag)::value;
        unsigned qf = qn != 0 ? 1u : 0u;
        bool more = qn != 0, have = cm != 0;
        if (__builtin_amdgcn_ballot_w64(drain ? (more | have) : (qn > keep)) != 0) do {
            unsigned m = cm;
            if (cm < qf) {      // fetch the next non-empty mask of MY queue
                const uint2 ne = *reinterpret_cast<const uint2*>(s_qb + raddr);
                m = ne.x; raddr = q_next(raddr); qn -= 1;
                qf = min((unsigned)qn, 1u);
                cbase = ne.y;
            }
            work_it += 1;
#ifdef SPHMI_STATS
            st_it += 1; st_lane += __builtin_popcountll(__builtin_amdgcn_ballot_w64(m != 0));
#endif
            if constexpr (kTwo) {
                // TWO neighbours per iteration, their four gathers in flight together; the pairs are still accumulated one after
                // the other, in mask order, so the sums are those of the one-pair loop bit for bit.
                const unsigned m1 = m & (m - 1);
                cm = m1 & (m1 - 1);
                if (m != 0) {
                    const bool two = m1 != 0;
                    const unsigned jr0 = rec_of(m, cbase);
                    const unsigned jr1 = two ? rec_of(m1, cbase) : jr0;
                    const V4 n0a = gather_packet(rs0, jr0, 0, T());
                    const V4 n1a = gather_packet(rs0, jr0, 1, T());
                    const V4 n0b = gather_packet(rs0, jr1, 0, T());
                    const V4 n1b = gather_packet(rs0, jr1, 1, T());
                    pair(jr0, n0a, n1a, plays_i(jr0));
                    if (two) pair(jr1, n0b, n1b, plays_i(jr1));
                }
            } else {
                cm = m & (m - 1);                                    // (0 stays 0)
                if (m != 0) {
                    const unsigned jr = rec_of(m, cbase);      // record size × the neighbour's index
                    const V4 n0 = gather_packet(rs0, jr, 0, T());
                    const V4 n1 = gather_packet(rs0, jr, 1, T());
                    pair(jr, n0, n1, plays_i(jr));
                }
            }
            more = (qf | cm) != 0u; have = false;
        } while (__builtin_amdgcn_ballot_w64(drain ? (more | have) : (qn > keep)) != 0);
    };
    // The pair loop software-pipelined by ONE address: the queue refill (LDS read), the bit walk and the record offset of the NEXT
    // neighbour are worked out while the two gathers of the current one are in flight (no further load in flight, one more
    // register) — the chain LDS → v_ffbl → offset → gather → arithmetic loses its first three links (profiles/HISTORY.md §4.6: 1.0673 → 1.0999e9
    // updates/s at C3).  (pv, pjr) = the pair this lane takes NEXT (valid flag, record offset); the state survives between the
    // bursts of the pair loop like the queue itself.
    // fp64 kernels: in the kernels of two waves per tile only.  Measured (µs per step, compiled-in / run-time models): 158 k
    // particles (two waves per tile) 360 / 462 with, 370 / 475 without; 470 k (one wave) 964 / 1070 with, 895 / 987 without; 1.06 M
    // 1960 / 2217 with, 1906 / 2071 without — one more register pair costs the run-time-model corrector its third wave per SIMD.
    constexpr bool kPipe = (sizeof(T) == 4 || WPT == 2) && !kTwoPairs && SPHMI_LDS_STAGE == 0;
    [[maybe_unused]] bool pv = false;
    [[maybe_unused]] unsigned pjr = 0;
    // (round 5: the loop test — a compare, a ballot and a branch — once per TWO iterations, SPHMI_LOOP_UNROLL: a burst may run one iteration longer than it
    // had to, which only moves a pair from the next burst into this one; the copy that rotated `cm` goes with it)
    // (the run-time-model kernels lose 5-12 % with it at 1 098 tiles: registers — compiled-in models only)
    constexpr bool kPairFetch64 = SPHMI_PAIR_FETCH64 != 0 && sizeof(T) == 8 && kHalf && MODEL >= 0 && SPHMI_DIAG == 0;
    [[maybe_unused]] const unsigned pf_c = (lane & 1) ? 16u : 0u, pf_self = a_r1 - (1u << kRecShift);
    auto run_pairs_piped = [&](const int keep, const bool drain) __attribute__((always_inline)) {
        unsigned qf = qn != 0 ? 1u : 0u;
        auto iteration = [&]() __attribute__((always_inline)) {
            work_it += 1;
#ifdef SPHMI_STATS
            st_it += 1; st_lane += __builtin_popcountll(__builtin_amdgcn_ballot_w64(pv));
#endif
            // 1. the gathers of the pair worked out an iteration (or a burst) ago
            const unsigned jr = pjr;
            const bool v = pv;
            V4 n0, n1;
            [[maybe_unused]] u32x4_t A, B, Cq, Dq;
            if constexpr (kPairFetch64) {
                // An fp64 record is 64 bytes = four 16-byte gathers per pair, and the texture path charges the instruction: 40 CU-cycles each for 64 scattered
                // lanes, 22 when adjacent lanes read the two halves of one 32-byte segment (tools/ubench/gather4.hip).  Four instructions still, but each fetches
                // one PACKET of 32 records: packets 0 and 1 of the even lanes' records, then of the odd lanes'; a lane holds half of every packet it needs and
                // its neighbour the other half — sixteen v_cndmask_b32_dpp hand them over (step 3).  1.06 M / 470 k / 159 k particles: 1 795 → 1 552, 806 → 691,
                // 293 → 253 µs per step (−14 %).  The fp32 version of the trade (two gathers of 34 cycles against eight exchanges) lost: profiles/r05_pair_gather_experiment.patch.
                const unsigned js = v ? jr : pf_self;                                                     // (a lane without a pair: any valid record)
                const unsigned jE = (unsigned)__builtin_amdgcn_mov_dpp((int)js, 0xA0, 0xF, 0xF, true) + pf_c;      // quad_perm [0, 0, 2, 2]
                const unsigned jO = (unsigned)__builtin_amdgcn_mov_dpp((int)js, 0xF5, 0xF, 0xF, true) + pf_c;      // quad_perm [1, 1, 3, 3]
                A = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)jE, 0, 0); B = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)(jE + 32u), 0, 0);
                Cq = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)jO, 0, 0); Dq = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)(jO + 32u), 0, 0);
            } else
#if SPHMI_DIAG == 5
            // DIAGNOSTIC BUILD (wrong results): every odd lane gathers the record its even neighbour gathers — what the texture path
            // charges when the two lanes of a pair address the same 32 bytes (the lane-pair design of profiles/HISTORY.md §4.6)
            if (v) { const unsigned js = (unsigned)__builtin_amdgcn_mov_dpp((int)jr, 0xA0, 0xF, 0xF, true);      // quad_perm [0, 0, 2, 2]
                     n0 = gather_packet(rs0, js, 0, T()); n1 = gather_packet(rs0, js, 1, T()); }
#elif SPHMI_DIAG == 2 || SPHMI_DIAG == 4
            // DIAGNOSTIC BUILD (wrong results): the arithmetic without the gathers — the floor set by the vector ALU
            if (v) { n0 = q0; n1 = q1; n0.x += __uint_as_float(jr) * T(1e-30); n0.y += T(0.003); n0.w = q0.w + T(1); }
#elif SPHMI_DIAG == 16
            // DIAGNOSTIC BUILD (wrong results): both packets from the LDS ring, at the slot the record index maps to
            if constexpr (kRing) { if (v) { const unsigned ro = ring_off0(jr); n0 = *reinterpret_cast<const V4*>(s_ringb + ro); n1 = *reinterpret_cast<const V4*>(s_ringb + ro + kRingOff1); } }
            else { if (v) { n0 = gather_packet(rs0, jr, 0, T()); n1 = gather_packet(rs0, jr, 1, T()); } }
#else
            { if (v) { n0 = gather_packet(rs0, jr, 0, T()); n1 = gather_packet(rs0, jr, 1, T()); } }
#endif
            // 2. while they fly: the next pair of this lane — refill when the mask is used up, lowest set bit, record offset
            unsigned m = cm;
            if (cm < qf) {
                const uint2 ne = *reinterpret_cast<const uint2*>(s_qb + raddr);
                m = ne.x; raddr = q_next(raddr); qn -= 1;
                if constexpr (sizeof(T) == 4) asm("v_min_u32 %0, %1, 1" : "=v"(qf) : "v"(qn));      // (one instruction; the compiler makes a compare + select of min(qn, 1))
                else qf = min((unsigned)qn, 1u);
                cbase = ne.y;
            }
            cm = m & (m - 1);
            pv = m != 0;
            // (meaningless without pv.  __builtin_ctz(0) is a POISON value in clang — llvm.cttz with is_zero_poison — not immediate
            // undefined behaviour: it is harmless as long as nothing consumes it, and every use of pjr sits under `if (v)`.  An inline
            // `v_ffbl_b32`, defined for 0, pins the LDS wait in front of the arithmetic and measured −0.5 %.)
            pjr = rec_of(m, cbase);
            // 3. the arithmetic
            if constexpr (kPairFetch64) {
                // even lane: packet 0 = { A, its neighbour's A }, packet 1 = { B, its neighbour's B };  odd lane: packet 0 = { its neighbour's C, C }, packet 1 = { its neighbour's D, D }
                unsigned w[16];
#define SPHMI_PF(d, x, y) "v_cndmask_b32_dpp %" #d ", %" #x ", %" #y ", vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                // low halves (bytes 0..15 of a packet), vcc = even lanes: even ? own first-record load : the even neighbour's second-record load
                asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_nop 1\n\t"
                             SPHMI_PF(0, 16, 8) SPHMI_PF(1, 17, 9) SPHMI_PF(2, 18, 10) SPHMI_PF(3, 19, 11)
                             SPHMI_PF(4, 20, 12) SPHMI_PF(5, 21, 13) SPHMI_PF(6, 22, 14) SPHMI_PF(7, 23, 15)
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[8]), "=&v"(w[9]), "=&v"(w[10]), "=&v"(w[11])
                             : "v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w),
                               "v"(Cq.x), "v"(Cq.y), "v"(Cq.z), "v"(Cq.w), "v"(Dq.x), "v"(Dq.y), "v"(Dq.z), "v"(Dq.w) : "vcc");
                // high halves (bytes 16..31), vcc = odd lanes: odd ? own second-record load : the odd neighbour's first-record load
                asm volatile("s_mov_b32 vcc_lo, 0xaaaaaaaa\n\ts_mov_b32 vcc_hi, 0xaaaaaaaa\n\ts_nop 1\n\t"
                             SPHMI_PF(0, 8, 16) SPHMI_PF(1, 9, 17) SPHMI_PF(2, 10, 18) SPHMI_PF(3, 11, 19)
                             SPHMI_PF(4, 12, 20) SPHMI_PF(5, 13, 21) SPHMI_PF(6, 14, 22) SPHMI_PF(7, 15, 23)
                             : "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]), "=&v"(w[12]), "=&v"(w[13]), "=&v"(w[14]), "=&v"(w[15])
                             : "v"(A.x), "v"(A.y), "v"(A.z), "v"(A.w), "v"(B.x), "v"(B.y), "v"(B.z), "v"(B.w),
                               "v"(Cq.x), "v"(Cq.y), "v"(Cq.z), "v"(Cq.w), "v"(Dq.x), "v"(Dq.y), "v"(Dq.z), "v"(Dq.w) : "vcc");
#undef SPHMI_PF
                auto dbl = [](unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
                n0.x = dbl(w[0], w[1]); n0.y = dbl(w[2], w[3]); n0.z = dbl(w[4], w[5]); n0.w = dbl(w[6], w[7]);
                n1.x = dbl(w[8], w[9]); n1.y = dbl(w[10], w[11]); n1.z = dbl(w[12], w[13]); n1.w = dbl(w[14], w[15]);
            }
            if (v) pair(jr, n0, n1, plays_i(jr));
        };
        if (__builtin_amdgcn_ballot_w64(drain ? (pv | ((qf | cm) != 0u)) : (qn > keep)) != 0) do {
            iteration();
            if constexpr (SPHMI_LOOP_UNROLL == 2 && sizeof(T) == 4) iteration();
        } while (__builtin_amdgcn_ballot_w64(drain ? (pv | ((qf | cm) != 0u)) : (qn > keep)) != 0);
    };
    // the two-pair loop with the addresses of the NEXT one or two neighbours worked out while the four gathers fly (compiled-in
    // model only: the run-time variant's two-pair kernels — MovingSquare2d — lose 3 % with it, registers): 2-D dam break 32.0 →
    // 30.2 µs per step, Dambreak3d Dp0.02 73.0 → 63.5
    constexpr bool kPipe2 = kTwoPairs && SPHMI_LDS_STAGE == 0 && MODEL >= 0;
    [[maybe_unused]] bool pv2 = false;
    [[maybe_unused]] unsigned pjr2 = 0;
    auto run_pairs_piped2 = [&](const int keep, const bool drain) __attribute__((always_inline)) {
        unsigned qf = qn != 0 ? 1u : 0u;
        if (__builtin_amdgcn_ballot_w64(drain ? (pv | ((qf | cm) != 0u)) : (qn > keep)) != 0) do {
            work_it += 1;
#ifdef SPHMI_STATS
            st_it += 1; st_lane += __builtin_popcountll(__builtin_amdgcn_ballot_w64(pv)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(pv2));
#endif
            const unsigned jr0 = pjr, jr1 = pjr2;
            const bool v0 = pv, v1 = pv2;
            V4 n0a, n1a, n0b, n1b;
            if (v0) {
                n0a = gather_packet(rs0, jr0, 0, T()); n1a = gather_packet(rs0, jr0, 1, T());
                n0b = gather_packet(rs0, jr1, 0, T()); n1b = gather_packet(rs0, jr1, 1, T());      // (jr1 = jr0 when there is no second one)
            }
            unsigned m = cm;
            if (cm < qf) {
                const uint2 ne = *reinterpret_cast<const uint2*>(s_qb + raddr);
                m = ne.x; raddr = q_next(raddr); qn -= 1;
                qf = min((unsigned)qn, 1u);
                cbase = ne.y;
            }
            const unsigned m1 = m & (m - 1);
            cm = m1 & (m1 - 1);
            pv = m != 0; pv2 = m1 != 0;
            pjr = rec_of(m, cbase);
            pjr2 = pv2 ? rec_of(m1, cbase) : pjr;
            if (v0) {
                pair(jr0, n0a, n1a, plays_i(jr0));
                if (v1) pair(jr1, n0b, n1b, plays_i(jr1));
            }
        } while (__builtin_amdgcn_ballot_w64(drain ? (pv | ((qf | cm) != 0u)) : (qn > keep)) != 0);
    };
    auto run_pairs = [&](const int keep, const bool drain) __attribute__((always_inline)) {
        if constexpr (kPipe2) run_pairs_piped2(keep, drain);
        else if constexpr (kPipe) run_pairs_piped(keep, drain);
        else run_pairs_plain(keep, drain, std::bool_constant<kTwoPairs>());
    };

    // ---- phase 1: one 64-candidate chunk against the 64 targets of the tile → one 64-bit accept mask
    // per lane (= per target).  fp32: the 64×64 matrix |c−t|² − H'² comes from the matrix cores
    // (v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain): A = candidates × (cx, cy, cz, |c|², 1, 0),
    // B = (−2tx, −2ty, −2tz, 1, |t|²−H'², 0) × targets.  A lane ends up with the 16 results of ITS target
    // column for 16 candidate rows per 32×32 block; their sign bits are shifted into a word with
    // v_alignbit (one op per result), and one v_permlane32_swap hands each target lane both halves.
    // The candidates are loaded lane-permuted so that bit b of the mask is candidate cb + b.
    const int bperm = kInterleave ? lane : (((lane >> 2) & 1) << 5) | ((lane >> 5) << 4) | (((lane >> 3) & 3) << 2) | (lane & 3);
    float B0[2], B1[2], B2[2], A2;
    if constexpr (kHalf) {
        // lanes l and l + 32 hold the same target: the operand layout without an exchange, one target block
        B0[0] = hl ? m2y : m2x; B1[0] = hl ? 1.0f : m2z; B2[0] = hl ? 0.0f : -thr;
        B0[1] = B1[1] = B2[1] = 0.0f;
    } else {
    B0[0] = m2x; B0[1] = m2y; swap_halves(B0[0], B0[1]);       // [T]: {k0: −2tx | k1: −2ty}
    B1[0] = m2z; B1[1] = 1.0f; swap_halves(B1[0], B1[1]);      //      {k2: −2tz | k3: 1}
    B2[0] = -thr; B2[1] = 0.0f; swap_halves(B2[0], B2[1]);     //      {k4: |t|²−H'² | k5: 0}
    }
    A2 = lane < 32 ? 1.0f : 0.0f;                               // candidates: {k4: 1 | k5: 0}
    // ---- half tiles: the same matrix from the f16-input matrix instruction ----------------------------------------------
    // v_mfma_f32_32x32x2_f32 runs on the VECTOR ALU's fp32 multipliers: every one of the six per chunk keeps the SIMD's vector issue busy
    // for ~80 cycles, for every wave on it (tools/ubench/poison_mix.hip: 1 per 32 double-rate instructions 95 -> 176 cycles per body, and
    // the waves next to it pay the same) — a quarter of a launch.  v_mfma_f32_32x32x16_f16 costs ~15 (same benchmark).
    // Tile-local coordinates scaled by sc = 16/H — or less, so that the reach of the tile stays below 192 and |c|^2 inside the half
    // range; every coordinate is split into hi + lo halves (products of halves are exact in the fp32 accumulator), so that
    // |c|^2 - 2 c.t + |t|^2 - (sc H)^2 (1 + eps)  is the sum of sixteen products:
    //   lanes 0-31  (k 0...7):  cxh m2xh  cxh m2xl  cxl m2xh  cxl m2xl   cyh m2yh  cyh m2yl  cyl m2yh  cyl m2yl      (m2 = -2t)
    //   lanes 32-63 (k 8...15): the same four for z,   cch 1   ccl 1   1 thh   1 thl                                (cc = |c|^2, th = |t|^2 - cut)
    // eps = 3e-4 + 2.5e-5 R^2 + 0.03/(sc H)^2  (R = reach of the tile in units of H) covers the residual of the splits (2^-20 relative
    // per value, both roundings towards zero), the fp32 accumulation and half-precision subnormals flushed by the matrix pipe (lo
    // parts below 6e-5 times a partner of at most 384).  An ordinary tile reaches 4-5 H: eps ~ 1e-3, 0.15 % more candidates for the
    // pair loop; a tile of spray that spans 100 H gets eps ~ 0.26.  The mask is a superset either way; the pair loop applies the exact cut.
    constexpr bool kF16 = kHalf && SPHMI_F16_SCAN != 0;
    [[maybe_unused]] u32x4s Bh = {0u, 0u, 0u, 0u};
    [[maybe_unused]] float hinv = 0.0f;
    if constexpr (kF16) {
        const float Hinv = __builtin_amdgcn_rsqf((float)P.H2);
        const float Rs = fast_sqrt(wave_max(owned ? tt : 0.0f)) * Hinv + 3.0f;            // wave-uniform
        hinv = Hinv * fminf(16.0f, 192.0f * fast_rcp(Rs));
        const float thr_s = (hinv * hinv) * (float)P.H2;                                  // (sc H)^2: 256 for ordinary tiles
        const float sx = owned ? txl * hinv : 0.0f, sy = owned ? tyl * hinv : 0.0f, sz = owned ? tzl * hinv : 0.0f;
        const float tts = sx * sx + sy * sy + sz * sz;
        const float eps16 = 3e-4f + 2.5e-5f * (Rs * Rs) + 0.03f * fast_rcp(thr_s);
        const float th = owned ? tts - thr_s * (1.0f + eps16) : 60000.0f;
        // (lanes l and l + 32 hold the same target: the lower one supplies k 0...7, the upper one k 8...15)
        const unsigned bx = split_hl(-2.0f * sx), by = split_hl(-2.0f * sy), bz = split_hl(-2.0f * sz), bt = split_hl(th), one2 = pk_h2(1.0f, 1.0f);
        Bh[0] = hl ? bz : bx; Bh[1] = Bh[0]; Bh[2] = hl ? one2 : by; Bh[3] = hl ? bt : by;
    }
    auto scan_chunk16 = [&](const int cb, const int HI, const V4& cpk) -> unsigned long long {
        const int c = cb + bperm;
        const bool cv = c < HI;
        // (a lane beyond the row's end: coordinates 0 and |c|^2 far beyond the cut — every sum stays positive)
        const float cx = cv ? (float)(cpk.x - ox) * hinv : 0.0f, cy = cv ? (float)(cpk.y - oy) * hinv : 0.0f, cz = cv ? (float)(cpk.z - oz) * hinv : 0.0f;
        const float cc = cv ? cx * cx + cy * cy + cz * cz : 60000.0f;
        unsigned X[4], Z[4];
        split_hh_ll(cx, X[0], X[1]); split_hh_ll(cy, X[2], X[3]);
        split_hh_ll(cz, Z[0], Z[1]); Z[2] = split_hl(cc); Z[3] = pk_h2(1.0f, 1.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) swap_halves(X[k], Z[k]);             // X: block of candidates 0...31, Z: block of candidates 32...63
        unsigned W = 0u;
#pragma unroll
        for (int C = 1; C >= 0; --C) {
            if (kInterleave && C == 1 && HI - cb <= 32) continue;       // (as in scan_chunk)
            u32x4s aw; aw[0] = C ? Z[0] : X[0]; aw[1] = C ? Z[1] : X[1]; aw[2] = C ? Z[2] : X[2]; aw[3] = C ? Z[3] : X[3];
            f32x16 d = {0};
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aw), __builtin_bit_cast(f16x8, Bh), d, 0, 0, 0);
#pragma unroll
            for (int r = 15; r >= 0; --r)
                W = __builtin_amdgcn_alignbit(W, __float_as_uint(d[r]), 31);
            __builtin_amdgcn_sched_barrier(0);
        }
        return (unsigned long long)W;
    };
    // (half tiles: through the buffer descriptor of the gathers — one address instruction instead of a compare, a select and 64-bit
    // pointer arithmetic; a lane beyond HI reads a record that phase 1 then discards (`cv`), one beyond the array reads zeros)
    auto chunk_packet = [&](const int cb, const int HI) -> V4 {
        const int c = cb + bperm;
        if constexpr (kHalf) return gather_packet(rs0, (unsigned)c << kRecShift, 0, T());
        else return P.src0[c < HI ? c : cb];
    };
    auto scan_chunk = [&](const int cb, const int HI, const V4& cpk) -> unsigned long long {
        const int c = cb + bperm;
        const bool cv = c < HI;
        float A0[2], A1[2];
        A0[0] = (float)(cpk.x - ox); A0[1] = (float)(cpk.y - oy); A1[0] = (float)(cpk.z - oz);
        A1[1] = cv ? A0[0] * A0[0] + A0[1] * A0[1] + A1[0] * A1[0] : 1e30f;
        swap_halves(A0[0], A0[1]);                              // [C]: {k0: cx | k1: cy}
        swap_halves(A1[0], A1[1]);                              //      {k2: cz | k3: |c|²}
        unsigned W[2] = {0u, 0u};
#pragma unroll
        for (int C = 1; C >= 0; --C) {
            // (interleaved half tiles: the last chunk of a row is half empty on average — when its candidates 32 … 63 do not exist their
            // block is not computed; its sixteen bits of every lane stay zero)
            if (kInterleave && C == 1 && HI - cb <= 32) continue;      // (C3: 890 -> 873 µs per step)
#pragma unroll
            for (int Tb = 0; Tb < (kHalf ? 1 : 2); ++Tb) {
                f32x16 d = {0};
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[C], B0[Tb], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[C], B1[Tb], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2[Tb], d, 0, 0, 0);
#pragma unroll
                for (int r = 15; r >= 0; --r)
                    W[Tb] = __builtin_amdgcn_alignbit(W[Tb], __float_as_uint(d[r]), 31);
                __builtin_amdgcn_sched_barrier(0);      // one 32×32 block in flight: 16 accumulator registers
            }
        }
        // (kHalf: W[0] IS this lane's share — its target against candidates cb + 32·hl … + 31)
        if constexpr (kHalf) return (unsigned long long)W[0];
        swap_halves(W[0], W[1]);
        return ((unsigned long long)W[1] << 32) | W[0];
    };

    int g0 = 0;                      // WPT > 1: chunks of the rows before this one, mod WPT (wave-uniform)
    auto row_offset = [&](const int seg) { return (D == 3) ? ((seg % 3) - 1) * P.nxp + ((seg / 3) - 1) * P.nxyp : (seg - 1) * P.nxp; };
    // Several waves per tile = a case too small to hide latency behind other waves: a lone wave pays the round trip of
    // every row's range look-up one after the other (≈0.8 µs each; tools/trace_small.py).  The waves of the tile need the
    // same ranges: with four or more waves they fetch them TOGETHER — row s by wave s % WPT, all requests in flight at once —
    // and share them in LDS (two-wave tiles keep their own look-ups: sharing costs them 6 %).
    constexpr bool kShareRanges = WPT >= 4 && !kHalf;
    __shared__ int2 s_rng[kShareRanges ? NSEG * kWave : 1];
    // Round 6: half tiles of several waves per half (every launch of four and eight waves per tile since round 4) lost that sharing with the full tiles it was
    // written for, and each wave was back to one exposed round trip per cell row — nine in 3-D — in launches whose whole life is 10–40 µs.  They now request the
    // ranges of ALL rows at once and park them in a private LDS column ([wave][row][lane]: no barrier, no other wave reads it): one round trip instead of NSEG.
    // (-DSPHMI_PREFETCH_RANGES=0: A/B builds)
    // (compiled-in models only: the run-time-model kernels of four waves per tile sit at 127 registers and pay 12–31 % for the eighteen more — 158 791 particles
    // Laminar 193 → 233 µs per step, 70 262 with PlanarShifting 94 → 123: profiles/r06_raw/prefetch_ranges_sizes_ab.txt)
    constexpr bool kPrefetchRanges = SPHMI_PREFETCH_RANGES != 0 && WPT >= 4 && kHalf && MODEL >= 0;
    __shared__ int2 s_rng_own[kPrefetchRanges ? WPT * TPB * NSEG * kWave : 1];
    [[maybe_unused]] int2* const s_rng_w = s_rng_own + (kPrefetchRanges ? wvb * NSEG * kWave + lane : 0);
    if constexpr (kShareRanges) {
#pragma unroll
        for (int k = 0; k < (NSEG + WPT - 1) / WPT; ++k) {
            const int seg = wv + k * WPT;
            if (seg < NSEG) {
                const int off = row_offset(seg);
                s_rng[seg * kWave + lane] = make_int2(valid ? P.cstart[key_a + off - 1] : 0, valid ? P.cstart[key_a + off + 2] : 0);
            }
        }
        __syncthreads();
    }
    [[maybe_unused]] int bal = 0;                       // SPHMI_BALANCE: pairs queued by the lower lane of this target so far − pairs queued by the upper one
    auto push_entry = [&](const unsigned bits, const int c0) {
        if (bits != 0) {
            *reinterpret_cast<uint2*>(s_qb + waddr) = make_uint2(bits, (unsigned)c0 << kRecShift); waddr = q_next(waddr); qn += 1;
        }
    };
    const int nseg = dead ? 0 : NSEG;                   // (a dead wave scans nothing: empty queues, zero sums, no stores — it only keeps the barriers whole)
    if constexpr (kPrefetchRanges) {
        if (!dead) {
            int2 rg[NSEG];
#pragma unroll
            for (int seg = 0; seg < NSEG; ++seg) {
                const int off = row_offset(seg);
                rg[seg] = make_int2(valid ? P.cstart[key_a + off - 1] : 0, valid ? P.cstart[key_a + off + 2] : 0);
            }
#pragma unroll
            for (int seg = 0; seg < NSEG; ++seg) s_rng_w[seg * kWave] = rg[seg];
        }
    }
#pragma unroll 1
    for (int seg = 0; seg < nseg; ++seg) {
        // the three x-adjacent cells of a row are one contiguous index range (x is the fastest sort axis)
        int lo_l, hi_l;
        if constexpr (kShareRanges) { const int2 rg = s_rng[seg * kWave + lane]; lo_l = rg.x; hi_l = rg.y; }
        else if constexpr (kPrefetchRanges) { const int2 rg = s_rng_w[seg * kWave]; lo_l = rg.x; hi_l = rg.y; }
        else {
            const int off = row_offset(seg);
            lo_l = valid ? P.cstart[key_a + off - 1] : 0;
            hi_l = valid ? P.cstart[key_a + off + 2] : 0;
        }
        // keys are sorted, cstart is monotone: the union over the tile is [lo(first), hi(last))
        const int LO = rl_i(lo_l, 0);
        const int HI = rl_i(hi_l, last_lane);
        // kInterleave: this lane's candidates of the chunk at cb are cb + 4·hl + 8g + k (g = 0 … 7, k = 0 … 3); how many of them sit below a
        // bound X relative to cb + 4·hl: below(X) = 4·(X >> 3) + min(X & 7, 4) (floor shift: exact for negative X too), and a chunk further on
        // it is 32 less — so the range of MY cells' bits is worked out once per row and moved by 32 per chunk
        [[maybe_unused]] int row_b0 = 0, row_b1 = 0;
        if constexpr (kInterleave) {
            auto below = [](const int X) -> int { return 4 * (X >> 3) + min(X & 7, 4); };
            row_b0 = below(lo_l - LO - 4 * hl); row_b1 = below(hi_l - LO - 4 * hl);
        }
        int first = 0;
        if constexpr (WPT > 1 && !kHalf) {
            // chunks dealt round-robin over ALL rows with four or more waves (counts differ by one at most); two-wave tiles keep
            // the rotation by row (kHalf: a wave scans every chunk of ITS 32 targets)
            if constexpr (WPT >= 4) first = (wv - g0) & (WPT - 1);
            else first = (wv + WPT - seg % WPT) % WPT;
            g0 = (g0 + (HI > LO ? (HI - LO + kWave - 1) / kWave : 0)) & (WPT - 1);
        }
        if constexpr (kHalf && kPar > 1) first = (par + seg) & (kPar - 1);       // (the waves of a half take alternate chunks, rotated by row)
#pragma unroll 1
        for (int cb = LO + first * kWave; cb < HI; cb += kHalf ? kWave * kPar : kWave * WPT) {
            // A tile of a sparse region (spray, a thin sheet) spans many cells: the union range of a row is then
            // mostly candidates that belong to NO lane's three cells.  Skip those chunks (two straggler tiles of
            // this kind doubled the launch time of the developed dam break: 1.10 → 0.6x ms).
            if (__builtin_amdgcn_ballot_w64((lo_l < cb + kWave) & (hi_l > cb)) == 0) continue;
            // room for the two entries of a chunk (its 32-candidate halves) in every lane's queue?
            if (__builtin_amdgcn_ballot_w64(qn > QCAP - 2) != 0) run_pairs(QCAP - 1 - kQueueSlack, false);
            unsigned long long m;
            {
                const V4 cpk = chunk_packet(cb, HI);
                if constexpr (kRing && SPHMI_RING_STAGE != 0) {
                    // what filling the ring costs: the chunk's second packets loaded with the first, both parked at the records' slots
                    // (a ring shared by the four waves of a block: each of them stages a quarter of the chunks)
                    if (SPHMI_RING_SHARED == 0 || ((cb >> 6) & 3) == wvb) {
                        const unsigned cr = (unsigned)(cb + bperm) << kRecShift;
                        const V4 cpk1 = gather_packet(rs0, cr, 1, T());
                        // (a lane beyond the array loads zeros — ρ = 0 would put an infinity into sums that the closing multiply by zero turns into NaN: its own record instead)
                        const bool in = cb + bperm < P.N;
                        V4 w0 = cpk, w1 = cpk1;
                        if (!in) { w0 = q0; w1 = q1; }
                        *reinterpret_cast<V4*>(s_ringb + ring_off0(cr)) = w0;
                        *reinterpret_cast<V4*>(s_ringb + ring_off0(cr) + kRingOff1) = w1;
                    }
                }
                if constexpr (kF16) m = scan_chunk16(cb, HI, cpk);
                else m = scan_chunk(cb, HI, cpk);
                // keep only the candidates of MY three cells of this row (the reference's stale cell list,
                // quirk Q1): bits [lo_l − cb, hi_l − cb) of the tile-wide mask
                if constexpr (kInterleave) {
                    // the bits of MY cells: [below(lo) − 32j, below(hi) − 32j) within 0 … 32 for the j-th chunk of the row (row_b0 / row_b1 above)
                    const int t = (cb - LO) >> 1;
                    const int b0 = min(max(row_b0 - t, 0), 32), b1 = min(max(row_b1 - t, 0), 32);
                    const int w = b1 - b0;
                    const unsigned rm = ((~0u) >> ((32 - w) & 31)) << (b0 & 31);
                    m = (w > 0) ? (unsigned long long)((unsigned)m & rm) : 0ull;
                } else if constexpr (kHalf) {
                    // (this lane's 32 candidates start at cb + 32·hl)
                    const int cbh = cb + 32 * hl;
                    const int b0 = max(lo_l - cbh, 0), b1 = min(hi_l - cbh, 32);
                    const int w = b1 - b0;
                    const unsigned rm = ((~0u) >> ((32 - w) & 31)) << (b0 & 31);
                    m = (w > 0) ? (unsigned long long)((unsigned)m & rm) : 0ull;
                } else {
                const int b0 = max(lo_l - cb, 0), b1 = min(hi_l - cb, 64);
                const int w = b1 - b0;
                const unsigned long long rm = ((~0ull) >> ((64 - w) & 63)) << (b0 & 63);
                m = (w > 0) ? (m & rm) : 0ull;
                }
            }
            work_ch += 1;
#ifdef SPHMI_STATS
            st_chunks += 1;
#endif
#if SPHMI_LDS_STAGE
            {
                // Chunk-synchronous LDS staging (the design BASELINE config 3 names; profiles/HISTORY.md §4.4 and profiles/r03_lds_stage_ablation.md
                // for why it is not what ships): every lane loads the record of ONE candidate, coalesced, and parks it in LDS; then
                // the wave walks the accept masks of THIS chunk — a lane reads its neighbour's two packets with two ds_read_b128 —
                // until the lane with the most accepted candidates in the chunk is done; lanes with fewer idle.  No mask queues.
                const int c = cb + lane;
                const bool cvs = c < HI;
                const V4 s0 = P.src0[cvs ? c : cb], s1 = P.src1[cvs ? c : cb];
                wave_sync();                                     // the readers of the previous chunk are done (one wave: program order)
                s_stage[2 * lane] = s0; s_stage[2 * lane + 1] = s1;
                wave_sync();
                unsigned long long mm = m;
                while (__builtin_amdgcn_ballot_w64(mm != 0ull) != 0) {
                    work_it += 1;
#ifdef SPHMI_STATS
                    st_it += 1; st_lane += __builtin_popcountll(__builtin_amdgcn_ballot_w64(mm != 0ull));
#endif
                    if (mm != 0ull) {
                        const int b = __builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        const V4 n0 = s_stage[2 * b], n1 = s_stage[2 * b + 1];
                        const unsigned jr = (unsigned)(cb + b) << kRecShift;
                        pair(jr, n0, n1, plays_i(jr));
                    }
                }
                continue;
            }
#endif
#if SPHMI_DIAG == 8
            // DIAGNOSTIC BUILD (wrong results): adjacent lanes — adjacent targets — walk the UNION of their accept masks in step and gather the same record
            m = (unsigned long long)((unsigned)m | (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)m, 0xB1, 0xF, 0xF, true));
#endif
            if constexpr (kInterleave && SPHMI_BALANCE != 0) {
                // Who takes which share.  The two lanes of a target get the candidates 8g + k and 8g + 4 + k of a chunk — a random half each, and a wave runs as many
                // pair iterations as its fullest LANE needs: 79.8 per half tile at rest with 85 % of the lane slots busy, where two equal halves would need 72.5
                // (tools/half_tile_balance_sim.py).  Every idle slot is texture-path time (profiles/HISTORY.md §4.9), so per chunk the lane that is BEHIND takes the larger of the two
                // words: whole words change hands (a 128-byte line stays with one lane, one queue entry per lane and chunk as before), the running difference `bal` =
                // pairs of the lower lane − pairs of the upper lane is kept identically on both.  Simulated: 74.5 iterations, 91 %.
                unsigned wl = (unsigned)m, wu = (unsigned)m;
                swap_halves(wl, wu);                                     // wl: the lower lane's word on both lanes, wu: the upper lane's
                const int cl = __builtin_popcount(wl), cu = __builtin_popcount(wu);
                const bool lower_takes_upper = (bal <= 0) != (cl >= cu);          // the lower lane is behind (or level) and its own word is the smaller one, or ahead and its own the larger
                const int d = lower_takes_upper ? cu - cl : cl - cu;
                bal += d;
                const bool take_upper = lower_takes_upper != (hl != 0);
                push_entry(take_upper ? wu : wl, cb + (take_upper ? 4 : 0));
            } else
            if constexpr (kInterleave) push_entry((unsigned)m, cb + 4 * hl);
            else if constexpr (kHalf) push_entry((unsigned)m, cb + 32 * hl);
            else { push_entry((unsigned)m, cb); push_entry((unsigned)(m >> 32), cb + 32); }
        }
    }
    run_pairs(0, true);
    if constexpr (kFast && kFastDiag) sum_d *= Kd_a;
    if constexpr (kFoldKv2) { const T k = P.Kv2 * P.Cfac; ax *= k; ay *= k; az *= k; sum_c *= P.Cfac; sum_d *= P.Cfac; }
    drho = rm_a * sum_c + sum_d;                        // continuity (src/SPHCellList.jl:289-291) + density diffusion
    if constexpr (kHalf) {
        // the two lanes of a target add their sums (lower half + upper half, in that order on both): every lane then holds the totals
        auto both = [](T v) -> T {
            if constexpr (sizeof(T) == 4) { float p = v, q = v; swap_halves(p, q); return p + q; }
            else {
                const unsigned long long u = (unsigned long long)__double_as_longlong(v);
                unsigned l0 = (unsigned)u, l1 = l0, h0 = (unsigned)(u >> 32), h1 = h0;
                swap_halves(l0, l1); swap_halves(h0, h1);
                return __longlong_as_double((long long)(((unsigned long long)h0 << 32) | l0)) + __longlong_as_double((long long)(((unsigned long long)h1 << 32) | l1));
            }
        };
        ax = both(ax); ay = both(ay); az = both(az); drho = both(drho);
        if (shift) { gcx = both(gcx); gcy = both(gcy); gcz = both(gcz); divr = both(divr); }
        if (MODEL < 0 && P.kout) { kgx = both(kgx); kgy = both(kgy); kgz = both(kgz); kw = both(kw); }
    }
#if SPHMI_DIAG != 0
    // (P.exact_cut is 0 at run time for the compiled-in model: the state stays sane — a select, not a multiply: a diagnostic loop may have produced NaN — and the
    // loop stays alive: the compiler cannot know)
    { const bool keep = P.exact_cut != 0; drho = keep ? drho : T(0); ax = keep ? ax : T(0); ay = keep ? ay : T(0); az = keep ? az : T(0); }
#endif
    // measured work of this tile (a pair-loop iteration ≈ 270, a chunk ≈ 475 vector-ALU cycles): the schedule of the rest
    // of the rebuild interval is rebuilt from it (Engine::reschedule)
    int tile_work = 9 * work_it + 16 * work_ch + 16;
    if (P.xcd_clock && lane == 0 && (kHalf || wv == 0)) {
        // one launch per rebuild interval is sampled: when does each XCD run out of tiles?  The engine moves the XCDs'
        // shares of the estimated cost towards equal finishing times at the next rebuild.
        atomicMax(&P.xcd_clock[blockIdx.x & 7], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        atomicMin(&P.xcd_clock[8], xcd_t0);
    }
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
#ifdef SPHMI_TRACE_WAVES
    // experiment: every wave of the tile leaves { start, end of its pair loop, hardware id, work } in a second table behind the
    // first (4 × tiles entries further): which SIMD of which unit ran it, and how the tile's waves compare
    if (lane == 0 && P.trace) {
        unsigned long long* W = P.trace + 4ull * ((unsigned long long)P.N / kWave + 2) + 4ull * ((unsigned long long)b * 8 + wv);
        W[0] = st_t0; W[1] = __builtin_amdgcn_s_memrealtime();
        W[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(blockIdx.x & 7) << 32);   // HW_ID | XCD
        W[3] = (unsigned long long)work_it | ((unsigned long long)work_ch << 32);
    }
#endif
    if (lane == 0 && wv == 0 && P.trace) {
        P.trace[4 * b] = st_entry;                                                                               // kernel entry of the wave
        P.trace[4 * b + 1] = st_t0;                                                                              // prologue done, scan starts
        P.trace[4 * b + 2] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)(blockIdx.x & 7) << 60);   // pair loop done (+ XCD of the block)
    }
#endif
#ifdef SPHMI_STATS
    if (lane == 0) {
        atomicAdd(&P.stats[0], st_it); atomicAdd(&P.stats[1], st_lane); atomicAdd(&P.stats[2], st_ref);
        atomicAdd(&P.stats[3], st_emp); atomicAdd(&P.stats[4], st_chunks); atomicAdd(&P.stats[5], 1ull);
    }
#endif
    if constexpr (kHalf) {
        // (a sampled launch: the tile's work = its slower wave × 2; the table was zeroed in front of the launch)
        tile_work *= WPT;
        if (P.tile_work && lane == 0 && !dead) atomicMax(&P.tile_work[b], tile_work);
        if constexpr (kPar > 1) {
            // two waves per half: the second hands its sums (already those of both lanes of a target) to the first; fixed order
            constexpr int kPartArrays = MODEL >= 0 ? 1 : 3;
            constexpr int kSet = (kPar - 1) * 64;                  // entries of one array: [wave of the half − 1][half][target of the half]
            __shared__ V4 s_hpart[kPartArrays * kSet];
            const int slot = half * 32 + (lane & 31);
            if (par > 0 && hl == 0) {
                const int w = (par - 1) * 64 + slot;
                V4 o; o.x = ax; o.y = ay; o.z = az; o.w = drho; s_hpart[w] = o;
                if (shift) { V4 g; g.x = gcx; g.y = gcy; g.z = gcz; g.w = divr; s_hpart[kSet + w] = g; }
                if (MODEL < 0 && P.kout) { V4 g; g.x = kgx; g.y = kgy; g.z = kgz; g.w = kw; s_hpart[2 * kSet + w] = g; }
            }
            __syncthreads();
            if (par > 0) return;
#pragma unroll
            for (int k = 0; k < kPar - 1; ++k) {
                const V4 o = s_hpart[k * 64 + slot];
                ax += o.x; ay += o.y; az += o.z; drho += o.w;
                if (shift) { const V4 g = s_hpart[kSet + k * 64 + slot]; gcx += g.x; gcy += g.y; gcz += g.z; divr += g.w; }
                if (MODEL < 0 && P.kout) { const V4 g = s_hpart[2 * kSet + k * 64 + slot]; kgx += g.x; kgy += g.y; kgz += g.z; kw += g.w; }
            }
        }
    } else if constexpr (WPT > 1) {
        // partial sums of waves 1 … WPT−1: { a, dρ/dt }, and for the run-time variant the shifting and kernel-output sums
        constexpr int kPartArrays = MODEL >= 0 ? 1 : 3;
        __shared__ V4 s_part_all[TPB * kPartArrays * (WPT - 1) * kWave];
        __shared__ int s_work_all[TPB * WPT];
        V4* const s_part = s_part_all + tib * kPartArrays * (WPT - 1) * kWave;
        int* const s_work = s_work_all + tib * WPT;
        // fixed summation order (wave 0 + wave 1 + …): results do not depend on which wave finishes first
        if (lane == 0) s_work[wv] = tile_work;
        if (wv > 0) {
            V4 o; o.x = ax; o.y = ay; o.z = az; o.w = drho; s_part[(wv - 1) * kWave + lane] = o;
            if (shift) { V4 g; g.x = gcx; g.y = gcy; g.z = gcz; g.w = divr; s_part[(WPT - 1 + wv - 1) * kWave + lane] = g; }
            if (MODEL < 0 && P.kout) { V4 g; g.x = kgx; g.y = kgy; g.z = kgz; g.w = kw; s_part[(2 * (WPT - 1) + wv - 1) * kWave + lane] = g; }
        }
        __syncthreads();
        if (wv > 0) return;
#ifdef SPHMI_TRACE_SYNC
        if (lane == 0 && P.trace) P.trace[4 * b + 1] = __builtin_amdgcn_s_memrealtime();     // experiment: when the slowest wave of the tile arrived
#endif
        // the workgroup holds its WPT wave slots until the slowest wave is done
#pragma unroll
        for (int k = 1; k < WPT; ++k) tile_work = max(tile_work, s_work[k]);
        tile_work *= WPT;
#pragma unroll
        for (int k = 0; k < WPT - 1; ++k) {
            const V4 o = s_part[k * kWave + lane];
            ax += o.x; ay += o.y; az += o.z; drho += o.w;
            if (shift) { const V4 g = s_part[(WPT - 1 + k) * kWave + lane]; gcx += g.x; gcy += g.y; gcz += g.z; divr += g.w; }
            if (MODEL < 0 && P.kout) { const V4 g = s_part[(2 * (WPT - 1) + k) * kWave + lane]; kgx += g.x; kgy += g.y; kgz += g.z; kw += g.w; }
        }
    }
    if (!kHalf && P.tile_work && lane == 0 && !dead) P.tile_work[b] = tile_work;
    // ---- epilogue ---------------------------------------------------------------------------
    // (the pre-test reading of the reduction slots is issued first: it comes from beyond the XCD's L2 and is needed last)

    const bool storer = owned && (!kHalf || hl == 0);      // (kHalf: both lanes of a target hold the results, the lower one stores)
    const uint8_t ty_a = ty_raw & 0x3F;
    const T gf = ty_a == 1 ? T(-1) : (ty_a == 3 ? T(1) : T(0));     // src/PreProcess.jl:78-87
    const T ml = fluid_a ? T(1) : T(0);
    if constexpr (PASS == PASS_FORCES_ONLY) {
        V4 o; o.x = ax; o.y = ay; o.z = az; o.w = drho;
        if (storer) P.accbuf[a] = o;
    } else if constexpr (PASS == PASS_PREDICTOR) {
        // HalfTimeStep (src/SPHCellList.jl:624-638) + LimitDensityAtBoundary! (SimulationEquations.jl:36-42)
        if constexpr (D == 3) az += P.g * gf; else ay += P.g * gf;
        V4 o0, o1;
        o0.x = xa + q1.x * step_dt2 * ml; o0.y = ya + q1.y * step_dt2 * ml; o0.z = za + q1.z * step_dt2 * ml;
        o1.x = q1.x + ax * step_dt2 * ml; o1.y = q1.y + ay * step_dt2 * ml; o1.z = q1.z + az * step_dt2 * ml;
        T rho_h = rho_a + drho * step_dt2;
        if (!fluid_a && rho_h < P.rho0) rho_h = P.rho0;
        o0.w = rho_h;
        o1.w = s_a;                         // ρⁿ·s travels with the half-step stream
        if (storer) { P.out0[a] = o0; P.out1[a] = o1; }
    } else {
        // LimitDensityAtBoundary!(Density) → DensityEpsi! → FullTimeStep
        // (src/SPHCellList.jl:794-798, 640-652; src/SimulationEquations.jl:28-33)
        V4 s0, s1;
        if constexpr (kEarlyEpilogueLoads) { s0 = pre_s0; s1 = pre_s1; } else { s0 = P.a0[ac]; s1 = P.a1[ac]; }
        // fp32 handles integrate ρ and x as double-floats (ForceParams::comp): state = what the record holds + the low word
        constexpr bool kComp = sizeof(T) == 4;
        const bool comp_on = kComp && P.comp != nullptr;
        V4 lo4; lo4.x = lo4.y = lo4.z = lo4.w = T(0);
        if constexpr (kEarlyEpilogueLoads) lo4 = pre_lo; else if (comp_on) lo4 = P.comp[ac];
        const T epsi = -(drho / rho_a) * step_dt;
        T rho_new;
        [[maybe_unused]] double rho_new_d = 0.0;
        if (comp_on) {
            double rho_n_d = (double)absT(s0.w) + (double)lo4.w;
            if (!fluid_a && rho_n_d < (double)P.rho0) rho_n_d = (double)P.rho0;
            rho_new_d = rho_n_d * ((2.0 - (double)epsi) / (2.0 + (double)epsi));
            rho_new = (T)rho_new_d;
        } else {
            T rho_n = absT(s0.w);
            if (!fluid_a && rho_n < P.rho0) rho_n = P.rho0;
            rho_new = rho_n * ((T(2) - epsi) / (T(2) + epsi));
        }
        if constexpr (D == 3) az += P.g * gf; else ay += P.g * gf;
        const T adx = ax * step_dt * ml, ady = ay * step_dt * ml, adz = az * step_dt * ml;
        V4 o0, o1, oa;
        o1.x = s1.x + adx; o1.y = s1.y + ady; o1.z = s1.z + adz;
        T sx = 0, sy = 0, sz = 0;
        if (shift) {
            // FullTimeStep with PlanarShifting, src/SPHCellList.jl:654-677 (A = 2, A_FST = 0, A_FSM = D)
            const T A_FSC = divr / T(D);
            if (!(A_FSC < T(0))) {
                const T k = -A_FSC * T(2) * P.h * fast_sqrt(o1.x * o1.x + o1.y * o1.y + o1.z * o1.z) * step_dt;
                sx = k * gcx; sy = k * gcy; sz = k * gcz;
            }
        }
        const T ix = (((o1.x + (o1.x - adx)) / T(2)) * step_dt + sx) * ml;
        const T iy = (((o1.y + (o1.y - ady)) / T(2)) * step_dt + sy) * ml;
        const T iz = (((o1.z + (o1.z - adz)) / T(2)) * step_dt + sz) * ml;
        if (comp_on) {
            const double xd = ((double)s0.x + (double)lo4.x) + (double)ix, yd = ((double)s0.y + (double)lo4.y) + (double)iy,
                         zd = ((double)s0.z + (double)lo4.z) + (double)iz;
            o0.x = (T)xd; o0.y = (T)yd; o0.z = (T)zd;
            lo4.x = (T)(xd - (double)o0.x); lo4.y = (T)(yd - (double)o0.y); lo4.z = (T)(zd - (double)o0.z);
            lo4.w = (T)(rho_new_d - (double)rho_new);
        } else { o0.x = s0.x + ix; o0.y = s0.y + iy; o0.z = s0.z + iz; }
        o0.w = fluid_a ? rho_new : -rho_new;
        o1.w = eos7<T>(rho_new, P.rho0, P.inv_rho0, P.Cbe);
        oa.x = ax; oa.y = ay; oa.z = az; oa.w = drho;
        if (storer) {
            if (MODEL < 0 && P.kout) { V4 ko; ko.x = kgx; ko.y = kgy; ko.z = kgz; ko.w = kw; P.kout[a] = ko; }
            P.out0[a] = o0; P.out1[a] = o1; P.accbuf[a] = oa;
            if (comp_on) P.comp[a] = lo4;
            // the sign of ρ carries the MotionLimiter flag, so ρ must stay positive
            if (!(rho_new > T(0))) atomicOr(&P.red[3], 1ull);
        }
        // reductions for the NEXT step: update_delta_x! (:706-724) and Δt (src/TimeStepping.jl:30-37)
        const T ddx = xa - o0.x, ddy = ya - o0.y, ddz = za - o0.z;     // Positionₙ⁺ − Position
        T disp2 = ddx * ddx + ddy * ddy + ddz * ddz;
        const T vr = o1.x * o0.x + o1.y * o0.y + o1.z * o0.z;
        const T rr = o0.x * o0.x + o0.y * o0.y + o0.z * o0.z;
        T vis = absT(P.h * vr / (rr + P.eta2));
        T a2 = ax * ax + ay * ay + az * az;
        if (!owned) { disp2 = T(0); vis = T(0); a2 = T(0); }      // tail lanes of the last tile
        disp2 = wave_max(disp2); vis = wave_max(vis); a2 = wave_max(a2);
        // every lane holds the three maxima: lanes 0, 1, 2 serve one slot each — ONE pre-test load and ONE atomic instruction
        // per wave instead of three dependent round trips to the coherence point (a device-scope load is served beyond the
        // XCD's L2; the epilogue of a lone wave: 5.3 → 4.1 µs, tools/trace_small.py)
        if (lane < 3) atomic_max_bits(&P.red[lane], lane == 0 ? disp2 : (lane == 1 ? vis : a2), /*pretest=*/WPT < 8 || SPHMI_SMALL_TRIMS == 0);
    }
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
    if (lane == 0 && wv == 0 && P.trace) {
        __builtin_amdgcn_s_waitcnt(0);                       // stores and atomics of the epilogue issued and returned
        P.trace[4 * b + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace sphmi
