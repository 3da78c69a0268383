// sphmi_engine.hip — host side of libsphmi.so: device state, the per-step launch sequence that
// replaces the reference's SimulationLoop (src/SPHCellList.jl:727-805) and the C ABI of
// include/sphmi.h.  gfx950 only; no CPU fallback: every entry point fails with SPHMI_ERR_DEVICE when
// HIP reports no usable device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sphmi.h"
#include "../../include/sphmi_internal.h"
#include "sphmi_kernels.h"
#include "sphmi_rebuild.h"

namespace sphmi {

struct EngineError : std::runtime_error {
    int status;
    EngineError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

#define HC(expr)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw EngineError(SPHMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static thread_local std::string g_create_error;

// Copies between device memory and PAGEABLE host memory — the caller's arrays, std::vector storage, stack variables — go
// through a page-locked bounce buffer that belongs to the engine.  Handed a pageable pointer, the runtime page-locks the
// caller's pages itself and remembers such pins by address; when the C library trims its heap and later hands the same
// addresses out again (or an array is unmapped and a new one lands on the same range) the next transfer into them runs on
// the stale mapping and the process dies with "Memory access fault by GPU … Write access to a read-only page" — seen about
// once per two runs of the GPU test suite (≈2 000 handle life cycles, host arrays allocated and freed all the time), in a
// download or an upload that had nothing wrong with it.  Memory the CALLER page-locked (sphmi_host_register) is copied
// directly, and so is everything the engine allocates with hipHostMalloc.
struct HostBounce {
    static constexpr size_t kBytes = size_t(16) << 20;
    static constexpr int kSlots = 8;                       // a ring of 2 MB pieces: the copy engine fills one while the host empties another
    static constexpr size_t kSlot = kBytes / kSlots;
    char* p = nullptr;
    hipEvent_t ev[kSlots] = {};
    HostBounce() = default;
    HostBounce(const HostBounce&) = delete;
    HostBounce& operator=(const HostBounce&) = delete;
    ~HostBounce() {
        if (p) (void)hipHostFree(p);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    }
    void ensure() {
        if (p) return;
        HC(hipHostMalloc((void**)&p, kBytes));
        for (auto& e : ev) HC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // device → pageable host; the data are in `dst` on return.  Piece k travels while piece k − 1 … is copied out of the ring
    // (1 M particles, every field into plain numpy arrays: 8.0 ms with one piece at a time, round 2's direct copies 3.3 ms).
    void d2h(void* dst, const void* src, size_t bytes, hipStream_t s) {
        ensure();
        const size_t pieces = (bytes + kSlot - 1) / kSlot;
        size_t out = 0;                                    // pieces already handed to the caller
        auto hand_out = [&]() {
            HC(hipEventSynchronize(ev[out % kSlots]));
            memcpy((char*)dst + out * kSlot, p + (out % kSlots) * kSlot, std::min(kSlot, bytes - out * kSlot));
            out += 1;
        };
        for (size_t k = 0; k < pieces; ++k) {
            if (k - out == (size_t)kSlots) hand_out();
            HC(hipMemcpyAsync(p + (k % kSlots) * kSlot, (const char*)src + k * kSlot, std::min(kSlot, bytes - k * kSlot), hipMemcpyDeviceToHost, s));
            HC(hipEventRecord(ev[k % kSlots], s));
            if (k > out) hand_out();                       // (keeps one piece in flight behind the one being copied out)
        }
        while (out < pieces) hand_out();
    }
    // pageable host → device; `src` may be reused on return (every piece has left the ring)
    void h2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
        ensure();
        const size_t pieces = (bytes + kSlot - 1) / kSlot;
        for (size_t k = 0; k < pieces; ++k) {
            if (k >= (size_t)kSlots) HC(hipEventSynchronize(ev[k % kSlots]));       // the piece that used this slot has reached the device
            const size_t n = std::min(kSlot, bytes - k * kSlot);
            memcpy(p + (k % kSlots) * kSlot, (const char*)src + k * kSlot, n);
            HC(hipMemcpyAsync((char*)dst + k * kSlot, p + (k % kSlots) * kSlot, n, hipMemcpyHostToDevice, s));
            HC(hipEventRecord(ev[k % kSlots], s));
        }
        HC(hipStreamSynchronize(s));
    }
};

// Phase labels follow the reference's TimerOutputs sections (src/SPHCellList.jl:748-800).
enum Phase { PH_TIMESTEP = 0, PH_REBUILD, PH_MDBC, PH_PASS1, PH_PASS2, PH_COUNT, PH_PASS1_EDGE, PH_PASS2_EDGE, PH_REBUILD_DEVICE };
static const char* kPhaseNames[PH_COUNT] = {
    "01 Update TimeStep", "02a Actual Calculate IndexCounter", "04 Apply MDBC before Half TimeStep",
    "05 First NeighborLoop", "08 Second NeighborLoop"};      // (05 holds the fused 06 / 07 half step, 08 the fused 09 / 10 / 11 full step)

// what the slab driver reads back after a batch of queued steps (Engine::dd_ctrl_sync)
struct sphmi_dd_control {
    int64_t steps_done;       // steps executed since dd_ctrl_init
    double total_time, last_dt, delta_x;
    int32_t need_rebuild;     // Δx ≥ h: rebuild (collectively), resume, keep queueing
    int32_t stop;             // loop bound reached
    int32_t error;            // 1: non-positive / NaN Δt, 2: non-positive density
    int32_t reserved;
};

struct EngineBase {
    sphmi_config cfg{};
    std::string err;
    int64_t iteration = 0, n_rebuilds = 0, index_counter = 0;
    double total_time = 0, last_dt = 0, delta_x = 0;
    bool uploaded = false;
    virtual ~EngineBase() {}
    virtual void upload(const void*, const void*, const void*, const void*, const uint8_t*, const int64_t*,
                        const uint64_t*, const void*) = 0;
    virtual void advance(double t_target, int64_t max_steps, sphmi_progress* out) = 0;
    virtual void download(void*, void*, void*, void*, void*, int64_t*, uint8_t*, uint64_t*, void*, int64_t*) = 0;
    virtual void download_begin(void*, void*, void*, void*, void*, int64_t*, uint8_t*, uint64_t*, void*, int64_t*) = 0;
    virtual void download_end() = 0;
    virtual void set_output_components(int c) = 0;
    virtual void host_register(void* p, size_t bytes) = 0;
    virtual void host_unregister(void* p) = 0;
    virtual void forces_once(int apply_mdbc, void* drhodt, void* acc) = 0;
    virtual void download_kernel_output(void* kernel, void* kernel_gradient) = 0;
    virtual void download_permutation(int64_t* prev_row) = 0;
    virtual void unique_cells(int64_t* out, int64_t cap, int64_t* n) = 0;
    virtual void timers(int32_t cap, const char** names, double* secs, int64_t* calls, int32_t* n) = 0;
    virtual void force_stats(int reset, double* avg_ms, int64_t* launches) = 0;
    virtual void device_ptrs(void** pk0, void** pk1, int64_t* n) = 0;
    virtual void reset_count() = 0;
    virtual int64_t owned_count() { return cfg.n_particles; }
    virtual void generate_dam_break_3d(double) { throw EngineError(SPHMI_ERR_STATE, "sphmi_generate_dam_break_3d: single-device handles only"); }
    virtual void set_motion(uint64_t group, double vel, double start, double dur, const double* dir) = 0;
};

template <class T>
struct Engine final : EngineBase {
    using V4 = typename Vec4<T>::type;
    int N = 0, D = 0;                  // N: live particles (≤ cap); the plain path keeps N == cap
    int cap = 0;                       // allocated particles
    bool own_stream = true;
    int* cellx_d = nullptr;
    hipStream_t stream = nullptr;
    // state sets: 0..2 rotate through the roles A (state n), H (half step), B (state n+1 / permute target)
    // particle records: rec[k] holds 2·cap packets, packet h of particle i at rec[k][2i + h]; pk0 / pk1 are the strided views
    V4* rec[3] = {};
    Half<V4> pk0[3], pk1[3];
    int iA = 0, iH = 1, iB = 2;
    V4 *acc[2] = {}, *ghost[2] = {};
    // fp32 handles: low words { x_lo, y_lo, z_lo, ρ_lo } of the double-float state (ForceParams::comp; $SPHMI_COMPENSATE=0: none —
    // the round-3 arithmetic, kept for the drift comparison of tests/test_config_scale_gpu.py).  fp64 handles: none.
    V4* comp[2] = {};
    V4* comp_cur() const { return comp[cur]; }
    uint8_t* type[2] = {};
    long long* id[2] = {};
    unsigned long long* grp[2] = {};
    unsigned long long* otag[2] = {};  // order tags (sphmi_rebuild.h, k_rankfix_tag): slab handles only
    int* prow[2] = {};                 // row of every particle at the last sphmi_download_permutation (the sort permutes it along)
    int* key[2] = {};
    int cur = 0;                       // which of the [2] copies is live
    int *slot = nullptr, *tmp_idx = nullptr, *perm = nullptr;
    int *count = nullptr, *cstart = nullptr, *tsum = nullptr;
    // tile schedules of the neighbour kernel: list 0 = interior (everything without a slab), list 1 = slab-edge tiles
    int *tile_cost[2] = {nullptr, nullptr}, *tile_order[2] = {nullptr, nullptr}, *tile_scan = nullptr, *tile_tsum = nullptr;
    int list_tiles[2] = {0, 0};        // tiles of each list (decides the waves per tile)
    int *part_d = nullptr, *part_h = nullptr;       // 2 × 16 ints: run starts and run lengths per XCD
    uint8_t* tile_cls = nullptr;
    unsigned long long* trace_d = nullptr;
    V4* kout_d = nullptr;              // StoreKernelOutput: { Σ∇W, ΣW } per particle
    MotionTable motions{};
    StepCtrl* ctrl_d = nullptr; StepCtrl* ctrl_h = nullptr;     // device-side step control (two blocks) + a pinned staging block (uploads; the slab driver's read-back)
    // the block of small control data on the device and its page-locked mirror: [2 × StepCtrl | 16 reduction slots | 8 rebuild counters / flags | 2 × 16 run-table words]
    static constexpr size_t kCtlRed = 256, kCtlMisc = kCtlRed + 16 * 8, kCtlPart = kCtlMisc + 16 * 4, kCtlBytes = kCtlPart + 32 * 4;
    // mDBC: the rows that carry a ghost node (k_permute appends them at every rebuild; their number is fixed at the upload), so that
    // k_mdbc launches one wave per ghost node instead of one per particle.  Slab engines (ghost copies come and go) look at every row.
    int* mdbc_list_d = nullptr; int mdbc_n_list = 0; bool mdbc_list_valid = false;
    int* mdbc_cnt_d() const { return misc_d + 8; }
    char* ctl_d = nullptr; char* ctl_m = nullptr;
    // Which of the two control blocks / two sets of reduction slots is current.  Plain handles take the control inside the
    // predictor (ForceParams::ctl_in): every queued step reads one block / set and writes the other, so both indices flip
    // per step at queue time; after a batch the control index is the last one written and the slot index the one the last
    // EXECUTED corrector filled.  Handles with mDBC, moving bodies or a slab keep index 0 (k_step_control works in place).
    int cpar = 0, rpar = 0;
    bool ghost_given = false;          // sphmi_upload handed in GhostPoints (they are permuted at every rebuild then)
    int fuse_ctrl = 1;                 // $SPHMI_FUSE_CTRL=0: a k_step_control launch per step for every handle (experiments)
    int same_cells = 1;                // $SPHMI_SAME_CELLS=0: every rebuild sorts, also when no particle changed its cell
    int64_t n_identity_rebuilds = 0;   // rebuilds that ended at the "nobody moved" test
    // Device-side rebuilds (round 4).  A cell-list rebuild of the host path waits for the device three times — the bounding box
    // sizes the grid, the run table sizes the force launches, the measured-work re-schedule sizes them again — which at the
    // reference's own example sizes (3 k … 160 k particles, a step of 30 … 150 µs) was 112–145 µs per rebuild, 10–19 % of a run.
    // Handles without a slab and with at most kSmallMaxTiles tiles keep the grid of their last HOST-side rebuild (its bounding box
    // + kStickySlack cell layers) and rebuild on it without asking the host anything: k_cell_count (flags a particle that left
    // the grid) → k_scan_single → k_scatter → k_rankfix → k_permute → k_tile_schedule_small, six launches, no round trip; the force
    // launches run with an upper-bound grid (8 × part_bound blocks) until the next batch boundary delivers the run table.
    // $SPHMI_DEVICE_REBUILD=0: the host path for every handle.
    int dev_rebuild = 1;
    int64_t n_device_rebuilds = 0, n_grid_overflows = 0; double dev_rebuild_secs = 0;
    bool count_clean = false;          // `count` is all zero (k_scan_single leaves it so)
    bool part_copy_queued = false;     // a copy of the run table into part_h is in flight: read it at the next synchronisation
    // The longest XCD run a device-side (re-)schedule may produce — the force launches use 8 × this many blocks until the host has
    // seen the run table.  One and a half times the longest run of the last table the host HAS seen: the clamp then hardly ever binds
    // (a bound of twice the mean run cut the long run of cheap bottom-plate tiles short and cost the 159 k-particle fp64 case 5 %),
    // and the launches of one batch carry half as many empty blocks again as they need.
    int part_seen = 0;                 // longest run of the last run table the host has read
    static constexpr int kExactGridFromTiles = 1024;
    int ntile_now() const { return (N + kWave - 1) / kWave; }
    int part_bound() const {
        const int nt = (N + kWave - 1) / kWave;
        return std::min(nt, std::max(part_seen + part_seen / 2 + 1, nt / 8 + 1));
    }
    bool device_rebuild_ok() const {
        return dev_rebuild && !dd_slab && have_grid && sticky_grid && (N + kWave - 1) / kWave <= kSmallMaxTiles;
    }
    bool sticky_wanted() const { return dev_rebuild && !dd_slab && (N + kWave - 1) / kWave <= kSmallMaxTiles; }
    bool sticky_grid = false;          // the grid was chosen with slack layers (by a host-side rebuild of a handle that qualifies)
    StepCtrl* ctrl_cur() const { return ctrl_d + cpar; }
    unsigned long long* red_cur() const { return red_d + 4 * rpar; }
    // (decided once per queued batch: before the first rebuild there is no tile schedule and the predictor launch is skipped —
    // nobody would take the decisions)
    // (plain handles take the control inside the predictor; mDBC handles inside k_mdbc, which runs first — $SPHMI_FUSE_MDBC=0: not)
    int fuse_mdbc = 1, fuse_mdbc_max_n = 1 << 30;
    bool fused_control() const {
        // (every wave of k_mdbc that has a ghost node repeats the decisions — ≈0.15 µs of fp64 arithmetic: worth the 6 µs launch
        // it replaces on the 2-D layouts, 37.5 against 40 µs per step; DucklingMDBC, 54 817 particles, loses 2 µs with it)
        const bool mdbc_ok = fuse_mdbc && cfg.mdbc == SPHMI_MDBC_SIMPLE && N <= fuse_mdbc_max_n;
        // (moving bodies: inside the first k_progress_motion of the step — MovingSquare2d 53 → 49 µs per step; with mDBC as well, the
        // one-thread launch stays)
        return fuse_ctrl && (cfg.mdbc == SPHMI_MDBC_NONE || (mdbc_ok && motions.n == 0)) && !dd_slab && have_grid && part_max[0] > 0;
    }
    bool batch_fused = false;
    static constexpr int kBatch = 32;  // most steps queued between two looks at the control flags
    double dx_rate = 0.0;              // Δx per step over the last batch: the next batch ends at the step expected to ask for a rebuild
    int part_max[2] = {0, 0};          // tiles in the longest XCD run of each list (grid = 8 × part_max blocks)
    int force_wpt = 0;                 // $SPHMI_WPT: waves per tile override (experiments)
    // Round 6, $SPHMI_EDGE_WPT_JOINT=1 (off): the slab-edge launch of an OVERLAPPED pass shares the chip with the interior launch of the same pass (side stream ‖
    // main stream), so the waves per tile of the edge list may follow the SUM of the two lists instead of the edge list's own length — 2 850 edge tiles of a C4 slab
    // run four waves per tile (72 ns of device time per tile against 59 for the interior's two-wave tiles, tools/slab_device_time.py).  Measured with every slab's
    // pass alone on the chip (tools/slab_pass_time.py, three runs): the MEAN pass of a slab gets 1-2 % shorter at 2 / 4 / 8 slabs, but the SLOWEST slab — the one a
    // step waits for — −1 % / +2.4 % / ±0: its few long-lived edge waves start late and stretch the tail.  Not a gain where it counts; left as a switch for the
    // first real multi-GPU run.
    bool edge_wpt_joint = false;
    bool edge_overlapped = false;      // set by dd_pass around an edge launch that runs beside the interior launch
    // XCD shares of the estimated tile cost, moved towards equal finishing times: one corrector launch per rebuild
    // interval records when each XCD ran out of tiles ($SPHMI_XCD_FEEDBACK=0 switches it off)
    int xcd_feedback = 1; bool xcd_sampled = false;
    // Round 6: the XCD-share feedback is for launches of a round or two of the wave slots.  Re-measured on the current kernels ($SPHMI_XCD_FEEDBACK=0/1 side by side,
    // profiles/r06_raw/xcd_feedback_*.txt): from 7 339 tiles up equal shares are FASTER — C3 bench +2.4 % over four interleaved pairs, developed flow +1.5 %, the
    // 470 k-particle instantiations −0.5 … −6.5 % per step — the finishing times it equalises are set by the tails of the runs, not by their work, and moving work to
    // match them lengthens the launch; below that (159 k particles and less) the two are level, with single instantiations ±2 … 4 % either way.  $SPHMI_XCD_FEEDBACK=2: every size.
    int xcd_feedback_max_tiles = 5000;
    bool xcd_feedback_for(int ntile) const { return xcd_feedback == 2 || (xcd_feedback == 1 && ntile < xcd_feedback_max_tiles); }
    // after a rebuild: 1 = the next eligible corrector launch measures the work of every tile and the schedule of the rest
    // of the interval is rebuilt from it; 2 = the one after that records the XCD finishing times; 0 = nothing pending
    int sched_state = 0; int resched = 1; int* tile_work_d = nullptr;
    int sched1_state = 0; int* tile_work1_d = nullptr; bool resched0_pending = false, resched1_pending = false;    // the same for the slab-edge list (its launch sits on the side stream)
    unsigned long long *xcd_clock_d = nullptr, *xcd_clock_h = nullptr;
    double xcd_w[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
    int xcd_segs = 0;                  // contiguous segments of the tile list per XCD run; 0 = by size ($SPHMI_XCD_SEGS overrides)
    // measured (both with the order from measured work): 108 tiles 4 > 2 > 1; 2481 … 6344 tiles 2 > 1 (+6 … +2 %);
    // 10512 tiles 2 = 1; 14032 / 16528 / 24676 tiles 1 > 2 (+3 / +5 / +7 %)
    // cost classes of the tile order (sphmi_rebuild.h): fine where the launch fits the chip at once, coarse where it does not
    // the end of every XCD run ordered again by sixteen classes of its own cost range (k_tile_order, `tail_permille`): launches of several rounds of the wave
    // slots only — where the launch fits the chip at once every tile starts at t = 0 and the classes are fine already.  $SPHMI_TAIL_SORT = per mille (0: off)
    int tail_sort = 200;
    int tail_sort_permille(int ntile) const { return ntile < classes_fine_below ? 0 : tail_sort; }
    int tile_classes(int ntile) const { return ntile < classes_fine_below ? SPHMI_TILE_CLASSES_ONE_ROUND : SPHMI_TILE_CLASSES; }
    // waves per tile by tile count (measured with the paired two-wave launches and sixteen classes, updates/s WPT 2 / WPT 1:
    // 2 481 tiles 8.03 / 7.16e8, 3 454: 8.47 / 8.22, 5 050: 9.19 / 9.02, 6 985: 9.62 / 9.80, 9 428: 9.76e8 / 1.022e9,
    // 11 689: 0.986 / 1.054e9, 16 527: 0.978 / 1.08e9)
    // Waves per tile by the number of tiles of the launch (tools/wpt_sweep.py; profiles/r03_raw/wpt_sweep.txt): 8 below `tiny`, 4 below
    // `small`, 2 below `medium`, else 1.  The crossovers depend on what a wave of the kernel holds: fp64 kernels and the run-time-model
    // kernels have fewer waves per SIMD to begin with and lose them to extra waves per tile earlier.
    //   fp32   512 / 1 024 / 6 000   (compiled-in models: measured crossovers ≈450 / ≈1 150, one and two waves level from 2.5 k tiles on.  Run-time
    //                                 models: Laminar would take 1 300 / 3 000 (−4 … −7 %), LaminarSPS + Complex loses 22 % with four waves at
    //                                 1 098 tiles — one table for all of them, so these stay)
    //   fp64   400 /   400 / 2 000   (four waves never win; every model gains: 159 k particles −7 … −22 %, DucklingMDBC 216 → 167 µs per step)
    int kWptMedium = -1;               // $SPHMI_WPT2_BELOW: overrides `medium` for every kernel
    int waves_per_tile(int ntile, bool generic) const {
        if (force_wpt > 0) return force_wpt;
        // Round 4: every kernel of two, four or eight waves per tile serves HALF tiles (sphmi_kernels.h, kHalf: a wave serves 32 targets with two
        // lanes each; with four / eight waves per tile the two / four waves of a half deal its chunks alternately).  Measured on the dam-break
        // lattice (µs per step, `tools/time_sizes.py`; profiles/r04_raw/half_tile_thresholds.txt):
        //   fp32  273 tiles: 8 → 46, 4 → 53;  381: 59 / 57;  489: 68 / 66;  2 482: 4 → 175, 2 → 180;  3 455: 226 / 224;  5 051: 325 / 299;
        //         two waves against ONE (rounds 1-4) at 6.3 k / 16.5 k / 44 k / 120 k tiles: 380 / 895 / 2 444 / 6 950 against 421 / 957 / 2 553 / 7 117
        //   fp64  273 tiles: 8 → 79, 4 → 71;  607: 4 → 107, 2 → 116;  881: 157 / 161;  1 098: 194 / 169;  16.5 k: 2 → 1 845, 1 → 1 924
        // The fp64 kernels of the run-time models (180 registers) keep one wave per tile above 2 000 tiles (+3 … +14 % with two there).
        int tiny = 330, small = 3000, medium = INT32_MAX;
        // (fp64: DucklingMDBC, 857 tiles of a k = 1.5 kernel, runs 110 µs per step with two waves and 120 with four: the crossover sits below the lattice's)
        if (sizeof(T) == 8) { tiny = 0; small = 800; if (generic) medium = 2000; }
        if (kWptMedium >= 0) medium = kWptMedium;
        return ntile < tiny ? 8 : (ntile < small ? 4 : (ntile < medium ? 2 : 1));
    }
    int classes_fine_below = 10000;    // $SPHMI_CLASSES_FINE_BELOW
    // domain decomposition: slab axis and the rank's cell-column range along it
    bool dd_slab = false; int dd_axis = 0; int64_t dd_col_lo = 0, dd_col_hi = 0; bool dd_has_lo = false, dd_has_hi = false;
    int64_t cell_cap = 0;
    int *bbox_d = nullptr, *misc_d = nullptr;              // misc: [0] nonempty, [1] scan total
    unsigned long long* red_d = nullptr;
    int *bbox_h = nullptr, *misc_h = nullptr;
    unsigned long long* red_h = nullptr;
    GridDesc grid{};
    bool have_grid = false, stepped = false, nonempty_pending = false;
    // timing
    struct Ev { hipEvent_t a, b; int phase; int weight; int bstep; };   // weight 0: not sampled; bstep: index of the step inside a queued batch (−1: none)
    int batch_step = -1;
    std::vector<Ev> ev_pool, ev_pending;
    double ph_secs[PH_COUNT] = {};
    int64_t ph_calls[PH_COUNT] = {};
    double force_ms = 0; int64_t force_launches = 0;
    int64_t ev_always_until = 2;       // every step is timed while iteration < this (start-up, and after a stats reset)

    explicit Engine(const sphmi_config& c) {
        cfg = c;
        N = cap = (int)c.n_particles;
        D = c.dims; out_comp = c.dims;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            throw EngineError(SPHMI_ERR_DEVICE, "no HIP device available (libsphmi has no CPU fallback)");
        if (c.device < 0 || c.device >= ndev) throw EngineError(SPHMI_ERR_ARGUMENT, "device ordinal out of range");
        HC(hipSetDevice(c.device));
        HC(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (const char* w = getenv("SPHMI_WPT")) { const int v = atoi(w); if (v == 1 || v == 2 || v == 4 || v == 8) force_wpt = v; }
        if (const char* w = getenv("SPHMI_EDGE_WPT_JOINT")) edge_wpt_joint = atoi(w) != 0;
        if (const char* w = getenv("SPHMI_TAIL_SORT")) { const int v = atoi(w); if (v >= 0 && v <= 1000) tail_sort = v; }
        if (const char* w = getenv("SPHMI_XCD_FEEDBACK")) xcd_feedback = atoi(w);
        xcd_trace = getenv("SPHMI_XCD_TRACE") != nullptr;          // one line per sampled launch on stderr: finishing time of each XCD ÷ mean, shares
        if (const char* w = getenv("SPHMI_TPB")) { const int v = atoi(w); if (v == 1 || v == 2 || v == 4) tpb = v; }
        if (const char* w = getenv("SPHMI_RESCHED")) resched = atoi(w);
        if (const char* w = getenv("SPHMI_TPB2")) tpb2 = atoi(w);
        if (const char* w = getenv("SPHMI_WPT2_BELOW")) kWptMedium = atoi(w);
        if (const char* w = getenv("SPHMI_CLASSES_FINE_BELOW")) classes_fine_below = atoi(w);
        HC(hipMalloc(&xcd_clock_d, 16 * 8)); HC(hipHostMalloc(&xcd_clock_h, 32 * 8));
        if (const char* w = getenv("SPHMI_XCD_SEGS")) { const int v = atoi(w); if (v >= 1 && v <= 4096) xcd_segs = v; }
        const size_t n = (size_t)N;
        for (int k = 0; k < 3; ++k) { HC(hipMalloc(&rec[k], 2 * n * sizeof(V4))); pk0[k] = Half<V4>(rec[k]); pk1[k] = Half<V4>(rec[k] + 1); }
        for (int k = 0; k < 2; ++k) {
            HC(hipMalloc(&acc[k], n * sizeof(V4)));
            HC(hipMalloc(&ghost[k], n * sizeof(V4))); HC(hipMemset(ghost[k], 0, n * sizeof(V4)));
            HC(hipMalloc(&type[k], n));
            HC(hipMalloc(&id[k], n * 8));
            HC(hipMalloc(&grp[k], n * 8));
            HC(hipMalloc(&key[k], n * 4));
            HC(hipMalloc(&prow[k], n * 4));
        }
        {
            const char* w = getenv("SPHMI_COMPENSATE");
            if (sizeof(T) == 4 && !(w && atoi(w) == 0))
                for (int k = 0; k < 2; ++k) { HC(hipMalloc(&comp[k], n * sizeof(V4))); HC(hipMemset(comp[k], 0, n * sizeof(V4))); }
        }
        HC(hipMalloc(&slot, n * 4)); HC(hipMalloc(&tmp_idx, n * 4)); HC(hipMalloc(&perm, n * 4));
        // $SPHMI_POISON=<byte>: the record sets and the accelerations start out filled with that byte (255 / 127: NaNs of either sign)
        // instead of whatever the allocation held — a row that is read before it was ever written then shows in the results of EVERY
        // run, not in one of a few (tests/test_multi_gpu.py::test_rows_never_written_are_never_read)
        if (const char* w = getenv("SPHMI_POISON")) {
            const int b = atoi(w) & 255;
            for (int k = 0; k < 3; ++k) HC(hipMemset(rec[k], b, 2 * n * sizeof(V4)));
            for (int k = 0; k < 2; ++k) HC(hipMemset(acc[k], b, n * sizeof(V4)));
        }
        const size_t nt = n / kWave + 2;
        for (int k = 0; k < 2; ++k) { HC(hipMalloc(&tile_cost[k], nt * 4)); HC(hipMalloc(&tile_order[k], 8 * nt * 4)); }
        HC(hipMalloc(&tile_scan, nt * 4)); HC(hipMalloc(&tile_cls, nt)); HC(hipMalloc(&tile_work_d, nt * 4)); HC(hipMalloc(&tile_work1_d, nt * 4));
        if (cfg.kernel_output == SPHMI_KOUT_STORE) { HC(hipMalloc(&kout_d, n * sizeof(V4))); HC(hipMemset(kout_d, 0, n * sizeof(V4))); }
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
        HC(hipMalloc(&trace_d, nt * 32 * 9)); HC(hipMemset(trace_d, 0, nt * 32 * 9));     // per tile 4 stamps + (-DSPHMI_TRACE_WAVES) 8 waves × 4
#endif
        HC(hipMalloc(&tile_tsum, (nt / kScanTile + 2) * 4));
        // ONE small block holds everything the host looks at after a batch of queued steps — the two control blocks, the two sets of
        // reduction slots, the rebuild's counters and flags, the run tables of the tile schedules — so that ONE device → host copy
        // per batch boundary brings all of it (round 3: two copies per boundary plus two per rebuild, ≈5 µs each on a 30 µs step)
        static_assert(2 * sizeof(StepCtrl) <= kCtlRed, "control block layout");
        HC(hipMalloc(&ctl_d, kCtlBytes)); HC(hipMemset(ctl_d, 0, kCtlBytes)); HC(hipHostMalloc(&ctl_m, kCtlBytes)); memset(ctl_m, 0, kCtlBytes);
        ctrl_d = (StepCtrl*)ctl_d; red_d = (unsigned long long*)(ctl_d + kCtlRed); misc_d = (int*)(ctl_d + kCtlMisc); part_d = (int*)(ctl_d + kCtlPart);
        red_h = (unsigned long long*)(ctl_m + kCtlRed); misc_h = (int*)(ctl_m + kCtlMisc); part_h = (int*)(ctl_m + kCtlPart);
        HC(hipHostMalloc(&ctrl_h, sizeof(StepCtrl)));
        if (const char* w = getenv("SPHMI_FUSE_CTRL")) fuse_ctrl = atoi(w);
        if (const char* w = getenv("SPHMI_FUSE_MDBC")) fuse_mdbc = atoi(w);
        if (const char* w = getenv("SPHMI_SAME_CELLS")) same_cells = atoi(w);
        if (const char* w = getenv("SPHMI_DEVICE_REBUILD")) dev_rebuild = atoi(w);
        HC(hipMalloc(&bbox_d, 8 * 4));
        HC(hipHostMalloc(&bbox_h, 8 * 4));
    }
    ~Engine() override {
        (void)hipSetDevice(cfg.device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& e : ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (auto& e : ev_pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (int k = 0; k < 3; ++k) (void)hipFree(rec[k]);
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(acc[k]); (void)hipFree(ghost[k]); (void)hipFree(type[k]); (void)hipFree(id[k]); (void)hipFree(otag[k]);
            (void)hipFree(grp[k]); (void)hipFree(key[k]); (void)hipFree(prow[k]); (void)hipFree(comp[k]);
        }
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); (void)hipEventDestroy(ev_packed); }
        for (auto& e : host_pinned) (void)hipHostUnregister(e.first);
        (void)hipFree(out_arena);
        (void)hipFree(slot); (void)hipFree(tmp_idx); (void)hipFree(perm);
        for (int k = 0; k < 2; ++k) { (void)hipFree(tile_cost[k]); (void)hipFree(tile_order[k]); }
        (void)hipFree(kout_d); (void)hipFree(tile_work_d); (void)hipFree(tile_work1_d); (void)hipFree(xcd_clock_d); (void)hipHostFree(xcd_clock_h);
        (void)hipFree(tile_scan); (void)hipFree(tile_cls); (void)hipFree(tile_tsum);
        (void)hipFree(count); (void)hipFree(cstart); (void)hipFree(tsum);
#if defined(SPHMI_STATS) || defined(SPHMI_TRACE)
        if (trace_d) {   // experiment build: start / end clock of every tile of the LAST launch → $SPHMI_TRACE_FILE
            const char* fn = getenv("SPHMI_TRACE_FILE");
            std::vector<unsigned long long> tr((size_t)(cap / kWave + 2) * 4 * 9);
            if (fn && hipMemcpy(tr.data(), trace_d, tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                FILE* f = fopen(fn, "wb");
                if (f) { fwrite(tr.data(), 8, tr.size(), f); fclose(f); }
            }
            (void)hipFree(trace_d);
        }
#endif
#ifdef SPHMI_STATS
        {   // experiment build: loop statistics of the neighbour kernel, summed over every launch
            unsigned long long st[16];
            if (hipMemcpy(st, red_d, sizeof st, hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[sphmi stats] wave-iterations %llu  lane-iterations %llu  refills %llu  empty refills %llu  chunks %llu  waves %llu\n",
                        st[8], st[9], st[10], st[11], st[12], st[13]);
        }
#endif
        (void)hipFree(ctl_d); (void)hipHostFree(ctl_m); (void)hipHostFree(ctrl_h); (void)hipFree(mdbc_list_d);
        (void)hipFree(bbox_d);
        (void)hipHostFree(bbox_h);
        (void)hipFree(cellx_d); (void)hipFree(uc_tsum);
        if (stream && own_stream) (void)hipStreamDestroy(stream);
    }

    // ---- timing helpers ---------------------------------------------------------------------
    // An event pair costs several µs of stream bubbles (27 µs per step on the 7 k-particle 2-D case whose
    // passes last 20 µs; 28 µs = 2 % per step at 1 M particles), so the phases are timed on one step in
    // ev_period and the sample is weighted accordingly; rebuilds are always timed.  The period follows the step: a timed step carries
    // ≈30 µs of bubbles (a ≈10 µs gap before and after every timed launch, profiles/r04_raw/window_curve.txt) — one in 8 is 0.4 % of a
    // 0.97 ms step but was 13 % of the 28 µs steps of the reference's 2-D examples: one in 32 below 400 µs per step, one in 64 below 100.
    static constexpr int kEvSample = 8;
    int ev_period = kEvSample;
    Ev begin_phase(int phase) {
        Ev e{};
        e.phase = phase; e.bstep = batch_step;
        e.weight = (phase == PH_REBUILD || phase == PH_REBUILD_DEVICE || iteration < ev_always_until) ? 1 : ((iteration % ev_period) == 0 ? ev_period : 0);
        if (e.weight == 0) return e;
        if (!ev_pool.empty()) { const int w = e.weight; e = ev_pool.back(); ev_pool.pop_back(); e.phase = phase; e.weight = w; e.bstep = batch_step; }
        else {
            // timing only: no system-scope fence when the event completes (the default writes back and invalidates the caches — a ≈10 µs
            // gap after every timed launch, and a cold L2 for the launch behind it); the values are read after a stream synchronisation
            HC(hipEventCreateWithFlags(&e.a, hipEventDisableSystemFence)); HC(hipEventCreateWithFlags(&e.b, hipEventDisableSystemFence));
        }
        HC(hipEventRecord(e.a, stream));
        return e;
    }
    void end_phase(Ev e) { if (e.weight == 0) return; HC(hipEventRecord(e.b, stream)); ev_pending.push_back(e); }
    // call only after a stream sync.  `executed`: steps of the queued batch that really ran — the launches of the
    // steps the device-side control cancelled returned at once and must not enter the averages
    void collect_events(int64_t executed = INT64_MAX) {
        for (auto& e : ev_pending) {
            float ms = 0;
            if (e.bstep >= 0 && e.bstep >= executed) { ev_pool.push_back(e); continue; }
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                // the edge-tile launch of a split pass is part of that pass: its time is added, it is not a call
                const bool edge = e.phase == PH_PASS1_EDGE || e.phase == PH_PASS2_EDGE;
                const int ph = e.phase == PH_PASS1_EDGE ? PH_PASS1 : (e.phase == PH_PASS2_EDGE ? PH_PASS2 : (e.phase == PH_REBUILD_DEVICE ? PH_REBUILD : e.phase));
                if (e.phase == PH_REBUILD_DEVICE) dev_rebuild_secs += ms * 1e-3;
                ph_secs[ph] += ms * 1e-3 * e.weight;
                ph_calls[ph] += edge ? 0 : e.weight;
                if (ph == PH_PASS1 || ph == PH_PASS2) { force_ms += ms * e.weight; force_launches += edge ? 0 : e.weight; }
            }
            ev_pool.push_back(e);
        }
        ev_pending.clear();
        if (force_launches > 0) {
            const double step_us = 2.0 * force_ms * 1e3 / (double)force_launches;
            ev_period = step_us >= 400.0 ? kEvSample : (step_us >= 100.0 ? 4 * kEvSample : 8 * kEvSample);
        }
    }

    static double decode(unsigned long long bits) {
        if constexpr (sizeof(T) == 4) { uint32_t u = (uint32_t)bits; float f; memcpy(&f, &u, 4); return (double)f; }
        else { double d; memcpy(&d, &bits, 8); return d; }
    }

    // The compiled-in model accumulates the accelerations in units of the viscosity constant Kv2 = 2·m₀·α·c₀·h and scales the
    // sums back once (sphmi_kernels.h, kFoldKv2): that needs Kv2 — IN THE KERNEL'S TYPE — to be a normal number with a finite
    // reciprocal.  A tiny α (or m₀·c₀·h product) underflows to 0 or a subnormal in fp32: such handles take the run-time variant,
    // which multiplies by Kv2 per pair.  One predicate for the kernel choice and for inv_Kv2 (round-3 advice).
    T kv2() const { return (T)(2.0 * cfg.m0 * cfg.alpha * cfg.c0 * cfg.h); }
    bool kv2_foldable() const { const T k = kv2(); return std::isnormal(k) && std::isfinite(T(1) / k) && std::isnormal(T(1) / k); }
    // … and evaluate Pressure! of a neighbour as ρ⁷·(Cb/γ/ρ₀⁷) − Cb/γ (fp32 kernels: ρ⁷ must stay finite up to 4ρ₀, the constant normal)
    T cbe7() const { return (T)(((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0) / std::pow(cfg.rho0, 7.0)); }
    // (fp32: ρ⁷ must neither overflow at 4 ρ₀ nor fall into the subnormals at ρ₀ / 4 — unit systems with ρ₀ ≈ 1e-6 — where ρ⁷·Cbe7 would lose the
    // neighbour pressure without a word; such handles take the run-time variant, which forms (ρ/ρ₀)⁷)
    bool eos_foldable() const { return sizeof(T) == 8 || (std::pow(4.0 * cfg.rho0, 7.0) < 1e37 && std::pow(0.25 * cfg.rho0, 7.0) > 1e-30 && std::isnormal(cbe7())); }
    bool compiled_in_model() const {
        // the models of the stock examples; a kernel cut off before it vanishes (k < 2) takes the variant with the per-pair cut
        return cfg.viscosity == SPHMI_VISC_ARTIFICIAL && cfg.density_diffusion == SPHMI_DDT_LINEAR && cfg.shifting == SPHMI_SHIFT_NONE &&
               cfg.kernel == SPHMI_KERNEL_WENDLAND_C2 && cfg.kernel_output == SPHMI_KOUT_NONE && kv2_foldable() && eos_foldable();
    }
    ForceParams<T> force_params(int src, int a, int out, double dt) const {
        ForceParams<T> P{};
        P.src0 = pk0[src]; P.src1 = pk1[src];
        P.a0 = pk0[a]; P.a1 = pk1[a];
        P.out0 = pk0[out]; P.out1 = pk1[out];
        P.accbuf = acc[cur]; P.comp = comp[cur];
        P.key = key[cur]; P.cstart = cstart; P.type = type[cur];
        P.red = red_cur(); P.stats = red_d + 8; P.ctrl = nullptr;
        P.N = N; P.nxp = grid.np[0]; P.nxyp = grid.np[0] * grid.np[1];
        P.dt = (T)dt; P.dt2 = (T)(dt * 0.5);
        P.H2 = (T)cfg.H2; P.h = (T)cfg.h; P.h_inv = (T)cfg.h_inv;
        P.Cgw = (T)(cfg.alphaD * 5.0 / (8.0 * cfg.h * cfg.h));
        P.Cfac = (T)(-8.0 * (cfg.alphaD * 5.0 / (8.0 * cfg.h * cfg.h))); P.nhinv_half = (T)(-0.5 * cfg.h_inv); P.big = (T)1099511627776.0;
        P.m0 = (T)cfg.m0;
        P.Kddt = (T)(cfg.delta_phi * cfg.h * cfg.c0 * cfg.m0);
        P.linfac = (T)(cfg.rho0 * cfg.g * ((1.0 / (cfg.Cb * cfg.gamma)) * cfg.rho0));
        // (η² = 0 is legal in the reference — src/SPHKernels.jl: η² ≥ 0 — whose pair loop never meets i == j; the gather kernel's accept masks DO hold the
        // self pair, every term of which is an exact zero as long as 1/(r² + η²) is finite: 1e-24·h² keeps it so and moves no other pair by an ulp, fp64 included)
        P.eta2 = (T)std::max(cfg.eta2, 1e-24 * cfg.h * cfg.h);
        P.Kv2 = kv2();
        P.inv_Kv2 = kv2_foldable() ? T(1) / P.Kv2 : T(1);
        P.visc = cfg.viscosity; P.ddt = cfg.density_diffusion; P.shift = cfg.shifting == SPHMI_SHIFT_PLANAR;
        P.exact_cut = !(cfg.H >= 2.0 * cfg.h);
        P.kernel = cfg.kernel; P.kout = kout_d;
        P.alphaD = (T)cfg.alphaD; P.tens_eps = (T)cfg.cubic_eps;
        {   // Wᵢⱼ(instance, dx) of tensile_correction: the reference passes dx where q is expected
            const double q = cfg.dx;
            const double w = cfg.kernel == SPHMI_KERNEL_CUBIC_SPLINE
                ? cfg.alphaD * ((q <= 1.0 ? 1.0 - 1.5 * q * q + 0.75 * q * q * q : 0.0) + (q > 1.0 && q <= 2.0 ? 0.25 * (2 - q) * (2 - q) * (2 - q) : 0.0))
                : 1.0;
            P.inv_Wdx = (T)(1.0 / w);
        }
        P.Klam = (T)(4.0 * cfg.m0 * cfg.nu0);
        P.sps_cs2 = (T)((cfg.smagorinsky_constant * cfg.dx) * (cfg.smagorinsky_constant * cfg.dx));
        P.sps_blin = (T)((2.0 / 3.0) * cfg.blin_constant * cfg.dx * cfg.dx);
        P.hyd_a = cfg.rho0 * cfg.g / cfg.Cb; P.hyd_b = cfg.rho0;
        P.rho0 = (T)cfg.rho0; P.inv_rho0 = (T)(1.0 / cfg.rho0);
        P.g = (T)cfg.g;
        P.Cbe = (T)((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0);
        P.Cbe7 = eos_foldable() ? cbe7() : T(0);
        return P;
    }

    // tiles per block of the one-wave-per-tile launches (3-D fp32 compiled-in model; $SPHMI_TPB = 1, 2 or 4 overrides).  Measured at
    // 1.06 M particles (kernel ms per launch): 1 → 0.5601, 2 → 0.5591, 4 → 0.5558; it is what lets ONE tile segment per XCD
    // (the L2-friendly schedule) run as fast as sixteen: 0.5562 against 0.5585 / 0.5558
    // (fp64 handles launch their one-wave tiles one per block: 32 KB of queues per four-tile block would leave the units five blocks where
    // registers allow four waves per SIMD anyway, and a block lives as long as its slowest tile — 470 k particles 921 → 895 µs per step, round 3)
    int tpb = 4;
    int tpb2 = 1;                      // two-wave tiles in pairs (workgroups of four waves); $SPHMI_TPB2=0 switches it off
    template <int PASS, int MODEL, int TPB> void launch_force_tpb(const ForceParams<T>& P, int list) {
        dim3 g(8 * ((part_max[list] + TPB - 1) / TPB)), b(kWave * TPB);
        hipLaunchKernelGGL((k_neighbor_force<T, 3, PASS, MODEL, 1, TPB>), g, b, 0, stream, P);
        HC(hipGetLastError());
    }
    template <int PASS, int MODEL, int WPT> void launch_force_wpt(const ForceParams<T>& P, int list) {
        if constexpr (WPT == 1 && MODEL >= 0 && sizeof(T) == 4) {
            if (D == 3 && tpb == 4) { launch_force_tpb<PASS, MODEL, 4>(P, list); return; }
            if (D == 3 && tpb == 2) { launch_force_tpb<PASS, MODEL, 2>(P, list); return; }
        }
        if constexpr (WPT == 2) {
            // two tiles of two waves per workgroup: four waves = one per SIMD of the compute unit ($SPHMI_TPB2=0: one tile per block)
            if (D == 3 && tpb2 == 4) {
                dim3 g2(8 * ((part_max[list] + 3) / 4)), b2(kWave * 8);
                hipLaunchKernelGGL((k_neighbor_force<T, 3, PASS, MODEL, 2, 4>), g2, b2, 0, stream, P);
                HC(hipGetLastError());
                return;
            }
            if (D == 3 && tpb2) {
                dim3 g2(8 * ((part_max[list] + 1) / 2)), b2(kWave * 4);
                hipLaunchKernelGGL((k_neighbor_force<T, 3, PASS, MODEL, 2, 2>), g2, b2, 0, stream, P);
                HC(hipGetLastError());
                return;
            }
        }
        dim3 g(8 * part_max[list]), b(kWave * WPT);
        if (D == 3) hipLaunchKernelGGL((k_neighbor_force<T, 3, PASS, MODEL, WPT>), g, b, 0, stream, P);
        else        hipLaunchKernelGGL((k_neighbor_force<T, 2, PASS, MODEL, WPT>), g, b, 0, stream, P);
        HC(hipGetLastError());
    }
    template <int PASS, int MODEL> void launch_force_model(ForceParams<T> P, int list) {
        if (part_max[list] == 0) return;
        P.order = tile_order[list]; P.part = part_d + 16 * list; P.trace = trace_d;
        // waves per tile: enough waves for several rounds of the 8192 wave slots of the chip (per list: the
        // slab-edge list of a domain-decomposed pass is much shorter than the interior list)
        // (fixed at the rebuild: the choice must not follow the measured run lengths; an overlapped edge launch counts the interior tiles it runs beside)
        const int ntile = (list == 1 && edge_overlapped && edge_wpt_joint) ? list_tiles[0] + list_tiles[1] : list_tiles[list];
        const int wpt = waves_per_tile(ntile, MODEL < 0);
        bool resched_after = false;
        // (the XCD finishing times may also be taken by the SECOND step of the batch whose first step measured the work — the schedule made
        // from that work is in place by then — so that a handle knows its XCD shares after ONE batch, not after its second rebuild interval)
        if (PASS == PASS_CORRECTOR && list == 0 && sched_state != 0 && wpt <= 2 && (batch_step == 0 || (batch_step == 1 && sched_state == 2))) {
            // (the first step of a batch executes unless the batch starts with a rebuild request; then nothing is
            // written and the sample is void: all-zero work keeps the schedule, all-zero ends are ignored)
            if (sched_state == 1 && resched) {
                HC(hipMemsetAsync(tile_work_d, 0, (size_t)((N + kWave - 1) / kWave) * 4, stream));
                P.tile_work = tile_work_d;
                resched_after = true;
                sched_state = xcd_feedback_for(ntile) ? 2 : 0;
            } else if (xcd_feedback_for(ntile)) {
                for (int k = 0; k < 8; ++k) xcd_clock_h[16 + k] = 0ull;
                for (int k = 8; k < 16; ++k) xcd_clock_h[16 + k] = ~0ull;
                HC(hipMemcpyAsync(xcd_clock_d, xcd_clock_h + 16, 16 * 8, hipMemcpyHostToDevice, stream));
                P.xcd_clock = xcd_clock_d;
                sched_state = 0; xcd_sampled = true;
            } else sched_state = 0;
        }
        if (resched_after) {
            if (wpt == 2) launch_force_wpt<PASS, MODEL, 2>(P, list); else launch_force_wpt<PASS, MODEL, 1>(P, list);
            resched0_pending = true;              // served before the next step is queued (outside the timed phase)
            return;
        }
        if (PASS == PASS_CORRECTOR && list == 1 && sched1_state == 1 && resched && wpt <= 2 && batch_step == 0) {
            // slab-edge list: measured here (side stream), re-ordered on the main stream once the streams have joined
            HC(hipMemsetAsync(tile_work1_d, 0, (size_t)((N + kWave - 1) / kWave) * 4, stream));
            P.tile_work = tile_work1_d;
            sched1_state = 0; resched1_pending = true;
        }
        if (wpt == 8) launch_force_wpt<PASS, MODEL, 8>(P, list);
        else if (wpt == 4) launch_force_wpt<PASS, MODEL, 4>(P, list);
        else if (wpt == 2) launch_force_wpt<PASS, MODEL, 2>(P, list);
        else launch_force_wpt<PASS, MODEL, 1>(P, list);
    }
    // list: 0 = interior tiles (all tiles when the handle has no slab), 1 = slab-edge tiles
    template <int PASS> void launch_force(const ForceParams<T>& P, int list = 0) {
        if (compiled_in_model()) {
            if (cfg.H >= 2.0 * cfg.h) launch_force_model<PASS, kModelDefault>(P, list);
            else launch_force_model<PASS, kModelDefaultCut>(P, list);
        }
        else      launch_force_model<PASS, kModelGeneric>(P, list);
    }

    void set_motion(uint64_t group, double vel, double start, double dur, const double* dir) override {
        if (!dir) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_motion: null direction");
        int m = 0;
        while (m < motions.n && motions.group[m] != group) ++m;
        if (m == 16) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_motion: more than 16 moving groups");
        if (m == motions.n) motions.n += 1;
        motions.group[m] = group; motions.vel[m] = vel; motions.start[m] = start; motions.dur[m] = dur;
        for (int d = 0; d < 3; ++d) motions.dir[m][d] = d < D ? dir[d] : 0.0;
    }
    // ProgressMotion (src/SPHCellList.jl:765,787) on state set A
    void progress_motion(double dt2, const StepCtrl* ctrl = nullptr, bool take_control = false) {
        if (motions.n == 0) return;
        MotionCtl mc{};
        if (take_control) {
            // reads control block `cpar` and slot set `rpar`, writes block cpar ^ 1, zeroes the set this step's corrector fills; the
            // caller flips both indices afterwards
            mc.ctl_in = ctrl_d + cpar; mc.ctl_out = ctrl_d + (cpar ^ 1);
            mc.red_in = red_d + 4 * rpar; mc.red_zero = red_d + 4 * (rpar ^ 1);
            mc.h = cfg.h; mc.c0 = cfg.c0; mc.CFL = cfg.CFL;
            ctrl = nullptr;
        }
        hipLaunchKernelGGL(k_progress_motion<T>, dim3((N + 255) / 256), dim3(256), 0, stream, pk0[iA], pk1[iA], type[cur],
                           (const unsigned long long*)grp[cur], N, motions, total_time, dt2, ctrl, comp[cur], mc);
        HC(hipGetLastError());
    }

    // finishing time of every XCD in the sampled launch (xcd_clock_h, copied and synchronised by the caller) → its share of the
    // estimated cost moves towards the speed it showed (damped; shares stay within ±20 % of an eighth)
    void apply_xcd_feedback() {
        xcd_sampled = false;
        const unsigned long long t0 = xcd_clock_h[8];
        double Tx[8], mean = 0; bool ok = t0 != ~0ull;
        for (int x = 0; x < 8 && ok; ++x) { ok = xcd_clock_h[x] > t0; Tx[x] = ok ? (double)(xcd_clock_h[x] - t0) : 0.0; mean += Tx[x] / 8; }
        if (ok && xcd_trace) {
            fprintf(stderr, "[sphmi xcd] it %lld tiles %d  T/mean:", (long long)iteration, list_tiles[0]);
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.3f", Tx[x] / mean);
            fprintf(stderr, "  shares:");
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.4f", xcd_w[x]);
            fprintf(stderr, "  mean %.1f us\n", mean * 0.01);
        }
        if (ok && xcd_feedback) {
            double sum = 0;
            for (int x = 0; x < 8; ++x) { xcd_w[x] *= std::sqrt(mean / Tx[x]); xcd_w[x] = std::min(0.15, std::max(0.10, xcd_w[x])); sum += xcd_w[x]; }
            for (int x = 0; x < 8; ++x) xcd_w[x] /= sum;
            // the schedule in use was cut with the OLD shares: cut it again from the same measured work (when that still describes the
            // order the particles are in) — otherwise the new shares would wait for the next rebuild that sorts, which a column at rest never has
            if (work_valid && resched && !dd_slab) resched0_pending = true;
        }
    }
    bool xcd_trace = false;
    bool work_valid = false;           // tile_work_d holds the measured work of the tiles in their PRESENT order

    // ---- UpdateNeighbors! -------------------------------------------------------------------
    void rebuild() {
        Ev ev = begin_phase(PH_REBUILD);
        const int nb256 = (N + 255) / 256;
        const int init[8] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0, 0};
        memcpy(bbox_h, init, sizeof(init));
        HC(hipMemcpyAsync(bbox_d, bbox_h, sizeof(init), hipMemcpyHostToDevice, stream));
        if (xcd_sampled) HC(hipMemcpyAsync(xcd_clock_h, xcd_clock_d, 16 * 8, hipMemcpyDeviceToHost, stream));   // read after the sync below
        const int nb_bbox = std::min(nb256, 512);
        // Nobody left the cell it was sorted into?  Then the reference's stable sort (:142) is the identity permutation: cell
        // list, order, tile schedule and its measured work stay what they are, and the rebuild ends at this round trip.  Every
        // sphmi_advance opens with a rebuild (Δx re-armed, :739) — short output intervals and step-by-step drivers hit this.
        const bool check_same = same_cells && have_grid && !dd_slab;
        const int* okey = check_same ? key[cur] : nullptr;
        if (D == 3) hipLaunchKernelGGL((k_cell_bbox<T, 3>), dim3(nb_bbox), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, bbox_d, okey, grid);
        else        hipLaunchKernelGGL((k_cell_bbox<T, 2>), dim3(nb_bbox), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, bbox_d, okey, grid);
        HC(hipGetLastError());
        HC(hipMemcpyAsync(bbox_h, bbox_d, 7 * 4, hipMemcpyDeviceToHost, stream));
        HC(hipStreamSynchronize(stream));
        if (xcd_sampled) apply_xcd_feedback();
        if (check_same && bbox_h[6] == 0) {
            n_rebuilds += 1; n_identity_rebuilds += 1;
            end_phase(ev);
            return;
        }
        int64_t ncell = 1;
        // (handles that rebuild on the device between host-side rebuilds: kStickySlack empty cell layers round the bounding box, so
        // that the grid outlives the motion of the next rebuild intervals)
        const int slack = sticky_wanted() ? kStickySlack : 0;
        for (int d = 0; d < 3; ++d) {
            if (d < D) {
                int64_t n = (int64_t)bbox_h[3 + d] - (int64_t)bbox_h[d] + 1;
                if (n <= 0 || n > (1ll << 30)) throw EngineError(SPHMI_ERR_NUMERIC, "non-finite particle position in cell hash");
                grid.gmin[d] = bbox_h[d] - slack;
                grid.np[d] = (int)n + 2 + 2 * slack;
            } else { grid.gmin[d] = 0; grid.np[d] = 1; }
            ncell *= grid.np[d];
        }
        sticky_grid = slack > 0;
        // (the reference keeps its cells in a Dict and has no such limit; a dense grid is what the kernels index.  Round 5: 2^30 cells by default — two
        // 4-GB arrays of a 288-GB device, allocated only when a run asks for them, and a scan of ≈3 ms per rebuild at that size: a run whose spray has
        // flown far goes on, slower, instead of ending with SPHMI_ERR_DOMAIN at 2^27)
        const int64_t budget = cfg.max_cells > 0 ? cfg.max_cells : (1ll << 30);
        if (ncell > budget) {
            char buf[200];
            snprintf(buf, sizeof(buf), "bounding cell grid %d x %d x %d = %lld cells exceeds max_cells = %lld "
                     "(a particle left the domain?)", grid.np[0], grid.np[1], grid.np[2], (long long)ncell, (long long)budget);
            throw EngineError(SPHMI_ERR_DOMAIN, buf);
        }
        grid.ncell = (int)ncell;
        if (ncell + 2 > cell_cap) {
            // The old arrays go first (at 2^30 cells the two generations do not have to fit side by side) and the handle never keeps a pointer to freed
            // memory: pointers and capacity are cleared BEFORE the allocations, so a failed hipMalloc — a busy device, several slabs or rank processes on
            // one GPU — leaves a handle whose next rebuild allocates again and whose destructor frees nothing twice (round-5 advisor finding).
            (void)hipFree(count); (void)hipFree(cstart); (void)hipFree(tsum);
            count = cstart = tsum = nullptr; cell_cap = 0;
            const int64_t want = (ncell + 2) + (ncell + 2) / 4;
            int *c0 = nullptr, *c1 = nullptr, *c2 = nullptr;
            const bool ok = hipMalloc(&c0, (size_t)want * 4) == hipSuccess && hipMalloc(&c1, (size_t)want * 4) == hipSuccess &&
                            hipMalloc(&c2, (size_t)((want + kScanTile - 1) / kScanTile + 1) * 4) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                (void)hipFree(c0); (void)hipFree(c1); (void)hipFree(c2);
                have_grid = false;
                char buf[260];
                snprintf(buf, sizeof(buf), "no device memory for a cell grid of %d x %d x %d = %lld cells (%.1f GB for the two cell arrays): a particle far from "
                         "the others inflates the bounding grid — set sphmi_config.max_cells to bound it", grid.np[0], grid.np[1], grid.np[2], (long long)ncell,
                         2.0 * 4.0 * (double)want / 1e9);
                throw EngineError(SPHMI_ERR_DOMAIN, buf);
            }
            count = c0; cstart = c1; tsum = c2; cell_cap = want;
        }
        HC(hipMemsetAsync(count, 0, (size_t)(ncell + 2) * 4, stream));
        HC(hipMemsetAsync(misc_d, 0, 8 * 4, stream));
        count_clean = false;
        CtrlPatch zero_only{}; zero_only.zero = mdbc_cnt_d();
        if (D == 3) hipLaunchKernelGGL((k_cell_count<T, 3>), dim3(nb256), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, grid, count, key[cur], slot, (int*)nullptr, zero_only);
        else        hipLaunchKernelGGL((k_cell_count<T, 2>), dim3(nb256), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, grid, count, key[cur], slot, (int*)nullptr, zero_only);
        // the scan runs over ncell + 1 entries: entry ncell is the graveyard of dead particles, so
        // cstart[ncell] = number of live particles and cstart[ncell + 1] = N
        const int nscan = (int)ncell + 1;
        const int ntiles = (nscan + kScanTile - 1) / kScanTile;
        hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(kScanThreads), 0, stream, count, cstart, nscan, tsum, misc_d);
        hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tsum, ntiles, misc_d + 1);
        hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(kScanThreads), 0, stream, cstart, nscan, tsum, misc_d + 1);
        hipLaunchKernelGGL(k_scatter, dim3(nb256), dim3(256), 0, stream, N, key[cur], slot, cstart, tmp_idx);
        if (otag[0]) hipLaunchKernelGGL(k_rankfix_tag, dim3(nb256), dim3(256), 0, stream, N, key[cur], slot, cstart, (int)ncell, tmp_idx, otag[cur], perm);
        else         hipLaunchKernelGGL(k_rankfix, dim3(nb256), dim3(256), 0, stream, N, key[cur], slot, cstart, (int)ncell, tmp_idx, perm);
        PermuteArgs<T> A{};
        const int nxt = cur ^ 1;
        A.pk0_in = pk0[iA]; A.pk1_in = pk1[iA]; A.acc_in = acc[cur]; A.ghost_in = ghost[cur];
        A.pk0_out = pk0[iB]; A.pk1_out = pk1[iB]; A.acc_out = acc[nxt]; A.ghost_out = ghost[nxt];
        A.type_in = type[cur]; A.type_out = type[nxt];
        A.id_in = id[cur]; A.id_out = id[nxt];
        A.grp_in = grp[cur]; A.grp_out = grp[nxt];
        A.key_in = key[cur]; A.key_out = key[nxt];
        A.tag_in = otag[cur]; A.tag_out = otag[nxt];
        A.prow_in = prow[cur]; A.prow_out = prow[nxt];
        A.comp_in = comp[cur]; A.comp_out = comp[nxt];
        // (GhostPoints travel with the sort whenever the caller uploaded some — the reference permutes the column with every other,
        // :142 — not only for mDBC handles: found as uninitialised GhostPoints in a download after an odd number of rebuilds)
        A.perm = perm; A.flags = nullptr; A.N = N; A.has_ghost = cfg.mdbc == SPHMI_MDBC_SIMPLE || ghost_given;
        A.mdbc_list = mdbc_list_ready() ? mdbc_list_d : nullptr; A.mdbc_cnt = mdbc_cnt_d(); mdbc_list_valid = A.mdbc_list != nullptr;
        hipLaunchKernelGGL(k_permute<T>, dim3(nb256), dim3(256), 0, stream, A);
        if (otag[0]) {
            for (int d = 0; d < D; ++d)
                if (grid.gmin[d] < -32000 || grid.gmin[d] + grid.np[d] > 32000)
                    throw EngineError(SPHMI_ERR_DOMAIN, "domain decomposition: cell coordinates beyond ±32000 (order tags hold 16 bits per axis)");
            hipLaunchKernelGGL(k_make_tags, dim3(nb256), dim3(256), 0, stream, N, key[nxt], cstart, grid, otag[nxt]);
        }
        HC(hipGetLastError());
        std::swap(iA, iB);
        cur = nxt;
        HC(hipMemcpyAsync(misc_h, misc_d, 2 * 4, hipMemcpyDeviceToHost, stream));
        nonempty_pending = true;
        {   // tile schedules of the neighbour kernel (sphmi_rebuild.h, "Tile schedule")
            const int ntile = (N + kWave - 1) / kWave;
            const int sb = (ntile + kScanTile - 1) / kScanTile;
            const int nlist = dd_slab ? 2 : 1;
            if (dd_slab) {
                const int lo_pad = dd_has_lo ? (int)(dd_col_lo - grid.gmin[dd_axis] + 1) : -1;
                const int hi_pad = dd_has_hi ? (int)(dd_col_hi - grid.gmin[dd_axis] + 1) : -1;
                hipLaunchKernelGGL(k_tile_class, dim3((ntile * 64 + 255) / 256), dim3(256), 0, stream, key[cur], type[cur], N, ntile,
                                   grid.np[0], grid.np[1], dd_axis, lo_pad, hi_pad, tile_cls);
            }
            hipLaunchKernelGGL(k_tile_cost, dim3((ntile + 255) / 256), dim3(256), 0, stream, key[cur], cstart,
                               dd_slab ? tile_cls : (const uint8_t*)nullptr, N, ntile, grid.np[0], grid.np[0] * grid.np[1], D, grid.ncell,
                               tile_cost[0], tile_cost[1]);
            for (int l = 0; l < nlist; ++l) {
                hipLaunchKernelGGL(k_scan_tile, dim3(sb), dim3(kScanThreads), 0, stream, tile_cost[l], tile_scan, ntile, tile_tsum, misc_d + 2);
                hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tile_tsum, sb, misc_d + 3);
                hipLaunchKernelGGL(k_scan_add, dim3(sb), dim3(kScanThreads), 0, stream, tile_scan, ntile, tile_tsum, misc_d + 3);
                // segments per XCD run (k_tile_order): ONE — a contiguous stretch of the domain per XCD, whose source rows stay in
                // that XCD's L2 (round 1 dealt 16 … 64 segments round-robin for a 2–3 % shorter launch tail at 2.85× the
                // algorithmic traffic; with four neighbouring tiles per block the gain is gone: 0.5562 vs 0.5558 ms at 1.06 M)
                const int nseg = l == 1 ? 1 : (xcd_segs > 0 ? xcd_segs : 1);
                XcdShares W{};
                // (with the measured re-schedule the shares belong to IT: the estimate-based order lives for one step)
                for (int x = 0; x < 8; ++x) W.cum[x + 1] = W.cum[x] + (float)(l == 0 && !resched ? xcd_w[x] : 0.125);
                W.cum[8] = 1.0f;
                hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, stream, tile_cost[l], tile_scan, ntile, tile_order[l], part_d + 16 * l, nseg, W, 0, tile_classes(ntile), tail_sort_permille(ntile), 16);
            }
            HC(hipGetLastError());
            HC(hipMemcpyAsync(part_h, part_d, 32 * 4, hipMemcpyDeviceToHost, stream));
            HC(hipStreamSynchronize(stream));
            part_copy_queued = false;
            for (int l = 0; l < 2; ++l) {
                part_max[l] = 0;
                list_tiles[l] = 0;
                if (l < nlist) for (int x = 0; x < 8; ++x) { part_max[l] = std::max(part_max[l], part_h[16 * l + 8 + x]); list_tiles[l] += part_h[16 * l + 8 + x]; }
            }
            part_seen = part_max[0];
        }
        have_grid = true;
        n_rebuilds += 1;
        sched_state = 1; sched1_state = dd_slab ? 1 : 0; resched0_pending = false; resched1_pending = false; work_valid = false;
        end_phase(ev);
    }

    // UpdateNeighbors! without the host (small handles, see `dev_rebuild`): the grid of the last host-side rebuild, six launches,
    // no synchronisation.  `ctrl`: the control block the next queued step reads — a particle outside the grid ends the batch there
    // with error 3 (Engine::advance then rebuilds on the host with a new grid; nothing was permuted, nothing stepped).
    SmallSched small_sched(const int* work, int keep_if_empty, StepCtrl* ctrl) const {
        SmallSched S{};
        const int ntile = (N + kWave - 1) / kWave;
        S.key = key[cur]; S.cstart = cstart; S.work = work;
        S.N = N; S.ntile = ntile; S.nxp = grid.np[0]; S.nxyp = grid.np[0] * grid.np[1]; S.D = D; S.ncell = grid.ncell;
        S.order = tile_order[0]; S.part = part_d;
        for (int x = 0; x < 8; ++x) S.W.cum[x + 1] = S.W.cum[x] + (float)((work || !resched) ? xcd_w[x] : 0.125);
        S.W.cum[8] = 1.0f;
        S.nclass = tile_classes(ntile); S.bound = part_bound(); S.keep_if_empty = keep_if_empty;
        S.flags = ctrl ? misc_d : nullptr; S.ctrl = ctrl;
        return S;
    }
    void rebuild_device(const CtrlPatch& patch) {
        Ev ev = begin_phase(PH_REBUILD_DEVICE);
        const int nb256 = (N + 255) / 256;
        const int ncell = grid.ncell;
        if (!count_clean) { HC(hipMemsetAsync(count, 0, (size_t)(ncell + 2) * 4, stream)); HC(hipMemsetAsync(misc_d, 0, 8 * 4, stream)); }
        if (D == 3) hipLaunchKernelGGL((k_cell_count<T, 3>), dim3(nb256), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, grid, count, key[cur], slot, misc_d, patch);
        else        hipLaunchKernelGGL((k_cell_count<T, 2>), dim3(nb256), dim3(256), 0, stream, pk0[iA], type[cur], N, (T)cfg.H_inv, grid, count, key[cur], slot, misc_d, patch);
        const int nscan = ncell + 1;
        if (nscan <= kScanSingleMax) {
            hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, count, cstart, nscan, misc_d);
            count_clean = true;
        } else {
            const int ntiles = (nscan + kScanTile - 1) / kScanTile;
            HC(hipMemsetAsync(misc_d, 0, 2 * 4, stream));
            hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(kScanThreads), 0, stream, count, cstart, nscan, tsum, misc_d);
            hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tsum, ntiles, misc_d + 1);
            hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(kScanThreads), 0, stream, cstart, nscan, tsum, misc_d + 1);
            count_clean = false;
        }
        hipLaunchKernelGGL(k_scatter, dim3(nb256), dim3(256), 0, stream, N, key[cur], slot, cstart, tmp_idx);
        hipLaunchKernelGGL(k_rankfix, dim3(nb256), dim3(256), 0, stream, N, key[cur], slot, cstart, ncell, tmp_idx, perm);
        PermuteArgs<T> A{};
        const int nxt = cur ^ 1;
        A.pk0_in = pk0[iA]; A.pk1_in = pk1[iA]; A.acc_in = acc[cur]; A.ghost_in = ghost[cur];
        A.pk0_out = pk0[iB]; A.pk1_out = pk1[iB]; A.acc_out = acc[nxt]; A.ghost_out = ghost[nxt];
        A.type_in = type[cur]; A.type_out = type[nxt];
        A.id_in = id[cur]; A.id_out = id[nxt];
        A.grp_in = grp[cur]; A.grp_out = grp[nxt];
        A.key_in = key[cur]; A.key_out = key[nxt];
        A.prow_in = prow[cur]; A.prow_out = prow[nxt];
        A.comp_in = comp[cur]; A.comp_out = comp[nxt];
        A.perm = perm; A.flags = misc_d; A.N = N; A.has_ghost = cfg.mdbc == SPHMI_MDBC_SIMPLE || ghost_given;
        A.mdbc_list = mdbc_list_ready() ? mdbc_list_d : nullptr; A.mdbc_cnt = mdbc_cnt_d(); mdbc_list_valid = A.mdbc_list != nullptr;
        hipLaunchKernelGGL(k_permute<T>, dim3(nb256), dim3(256), 0, stream, A);
        std::swap(iA, iB);
        cur = nxt;
        nonempty_pending = true;               // (misc_d[0] and the run table travel with the copy of the control block at the next batch boundary)
        hipLaunchKernelGGL(k_tile_schedule_small, dim3(8), dim3(1024), 0, stream, small_sched(nullptr, 0, ctrl_cur()));
        HC(hipGetLastError());
        part_copy_queued = true;
        part_max[0] = part_bound(); part_max[1] = 0;
        list_tiles[0] = (N + kWave - 1) / kWave; list_tiles[1] = 0;
        n_rebuilds += 1; n_device_rebuilds += 1;
        sched_state = 1; sched1_state = 0; resched0_pending = false; resched1_pending = false; work_valid = false;
        end_phase(ev);
    }
    // The rebuild in front of the next queued step, with what the host has to tell the control block `c` (mode 1: all of it, a new
    // call; mode 2: "served").  Device-side: the rebuild's first launch writes it; host-side: an upload, then the rebuild.
    void rebuild_any(const StepCtrl& c, int mode) {
        if (device_rebuild_ok()) {
            CtrlPatch p{}; p.dst = ctrl_cur(); p.mode = mode; if (mode == 1) p.value = c;
            p.zero = mdbc_cnt_d();
            rebuild_device(p);
            return;
        }
        *ctrl_h = c;
        HC(hipMemcpyAsync(ctrl_cur(), ctrl_h, sizeof(StepCtrl), hipMemcpyHostToDevice, stream));
        rebuild();
    }

    // The schedule of list 0 from the MEASURED work of every tile (the sampled corrector launch just queued): same
    // segments, classes and XCD shares as k_tile_order of the rebuild, true costs instead of candidate counts.
    void reschedule_from_work(int list) {
        if (list == 0) work_valid = true;
        if (list == 0 && device_rebuild_ok()) {
            // small handles: one launch, and nobody waits for the table (the launches keep their upper-bound grid until the next
            // batch boundary has seen it)
            hipLaunchKernelGGL(k_tile_schedule_small, dim3(8), dim3(1024), 0, stream, small_sched(tile_work_d, 1, nullptr));
            HC(hipGetLastError());
            part_copy_queued = true;
            part_max[0] = part_bound();
            if (ntile_now() >= kExactGridFromTiles) {
                // From ≈64 k particles on the exact grid is worth one round trip per rebuild interval: an upper-bound grid (half as many
                // blocks again, all of them returning at once) made the fp64 corrector of the 159 k-particle case 17 % longer for the
                // rest of the batch (341 against 331 µs per step over 200 steps; the small cases gain more from not waiting)
                HC(hipMemcpyAsync(ctl_m + kCtlPart, ctl_d + kCtlPart, 16 * 4, hipMemcpyDeviceToHost, stream));
                HC(hipStreamSynchronize(stream));
                part_copy_queued = false;
                int m = 0;
                for (int x = 0; x < 8; ++x) m = std::max(m, part_h[8 + x]);
                if (m > 0) { part_max[0] = std::min(m, part_bound()); part_seen = part_max[0]; }
            }
            return;
        }
        const int ntile = (N + kWave - 1) / kWave;
        const int sb = (ntile + kScanTile - 1) / kScanTile;
        int* work = list == 0 ? tile_work_d : tile_work1_d;
        hipLaunchKernelGGL(k_scan_tile, dim3(sb), dim3(kScanThreads), 0, stream, work, tile_scan, ntile, tile_tsum, misc_d + 2);
        hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tile_tsum, sb, misc_d + 3);
        hipLaunchKernelGGL(k_scan_add, dim3(sb), dim3(kScanThreads), 0, stream, tile_scan, ntile, tile_tsum, misc_d + 3);
        const int nseg = list == 1 ? 1 : (xcd_segs > 0 ? xcd_segs : 1);
        XcdShares W{};
        for (int x = 0; x < 8; ++x) W.cum[x + 1] = W.cum[x] + (float)(list == 0 ? xcd_w[x] : 0.125);
        W.cum[8] = 1.0f;
        hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(1024), 0, stream, work, tile_scan, ntile, tile_order[list], part_d + 16 * list, nseg, W, 1, tile_classes(ntile), tail_sort_permille(ntile), 16);
        HC(hipGetLastError());
        // the grid of the following launches needs the longest run: one short host round trip per rebuild interval
        // (≈30 µs every ≈40 steps); a sample whose step was cancelled left the table as it was
        HC(hipMemcpyAsync(part_h, part_d, 32 * 4, hipMemcpyDeviceToHost, stream));
        HC(hipStreamSynchronize(stream));
        part_copy_queued = false;
        int m = 0;
        for (int x = 0; x < 8; ++x) m = std::max(m, part_h[16 * list + 8 + x]);
        if (m > 0) part_max[list] = m;
        if (list == 0) part_seen = part_max[0];
    }

    // take_control: the kernel takes the decisions of the step itself (MdbcParams::ctl_in): reads control block `cpar` and slot set
    // `rpar`, writes block cpar ^ 1; its flag goes to the set the coming corrector fills.  The caller flips both indices afterwards.
    static constexpr int kMdbcGroupFrom = 4096;
    const int mdbc_group_env = getenv("SPHMI_MDBC_GROUP") ? atoi(getenv("SPHMI_MDBC_GROUP")) : -1;
    bool mdbc_list_ready() {
        if (cfg.mdbc != SPHMI_MDBC_SIMPLE || dd_slab || mdbc_n_list <= 0) return false;
        if (!mdbc_list_d) HC(hipMalloc(&mdbc_list_d, (size_t)mdbc_n_list * 4));
        return true;
    }
    void run_mdbc(const StepCtrl* ctrl = nullptr, bool take_control = false) {
        Ev ev = begin_phase(PH_MDBC);
        MdbcParams<T> M{};
        M.ctrl = ctrl;
        M.pk0 = pk0[iA]; M.comp = comp[cur]; M.ghost = ghost[cur]; M.type = type[cur]; M.cstart = cstart; M.g = grid; M.red = red_cur(); M.N = N;
        if (take_control) {
            M.ctrl = nullptr;
            M.ctl_in = ctrl_d + cpar; M.ctl_out = ctrl_d + (cpar ^ 1);
            M.red_in = red_d + 4 * rpar; M.red = red_d + 4 * (rpar ^ 1);
            M.ctl_h = cfg.h; M.ctl_c0 = cfg.c0; M.ctl_CFL = cfg.CFL;
        }
        M.H_inv = (T)cfg.H_inv; M.H2 = cfg.H2; M.h_inv = cfg.h_inv; M.h = cfg.h;
        M.alphaD = cfg.alphaD; M.m0 = cfg.m0; M.rho0 = cfg.rho0; M.eta2 = cfg.eta2; M.kernel = cfg.kernel;
        M.gfac = cfg.alphaD * 5.0 / (8.0 * cfg.h * cfg.h);
        const bool listed = mdbc_list_valid && !dd_slab;
        if (listed) { M.list = mdbc_list_d; M.n_list = mdbc_n_list; }
        // sixteen lanes per node from kMdbcGroupFrom nodes on (enough waves to fill the chip four nodes at a time; below, the launch is
        // latency-bound and the shorter wave of the one-node kernel wins); SPHMI_MDBC_GROUP=0 / 1 forces either
        const bool grouped = listed && (mdbc_group_env >= 0 ? mdbc_group_env != 0 : mdbc_n_list >= kMdbcGroupFrom);
        if (grouped) {
            dim3 g((mdbc_n_list + 15) / 16), b(256);
            if (D == 3) hipLaunchKernelGGL((k_mdbc_group<T, 3>), g, b, 0, stream, M);
            else        hipLaunchKernelGGL((k_mdbc_group<T, 2>), g, b, 0, stream, M);
        } else {
            dim3 g(((listed ? mdbc_n_list : N) + 3) / 4), b(256);       // one wave per ghost node (per particle without the list), four per block
            if (D == 3) hipLaunchKernelGGL((k_mdbc<T, 3>), g, b, 0, stream, M);
            else        hipLaunchKernelGGL((k_mdbc<T, 2>), g, b, 0, stream, M);
        }
        HC(hipGetLastError());
        end_phase(ev);
    }

    void sync_and_collect(const StepCtrl* batch_ctrl = nullptr, int64_t steps_before = 0, bool mirror_fresh = false) {
        // (the counters and the run table of a device-side rebuild live in the control block: fetch them unless the caller just did)
        if (!mirror_fresh && (nonempty_pending || part_copy_queued))
            HC(hipMemcpyAsync(ctl_m + kCtlMisc, ctl_d + kCtlMisc, kCtlBytes - kCtlMisc, hipMemcpyDeviceToHost, stream));
        HC(hipStreamSynchronize(stream));
        collect_events(batch_ctrl ? batch_ctrl->steps_done - steps_before : INT64_MAX);
        if (nonempty_pending) { index_counter = (int64_t)misc_h[0] + 1; nonempty_pending = false; }
        if (part_copy_queued) {
            // the run table of the last device-side (re-)schedule has arrived: the exact grid from here on
            part_copy_queued = false;
            int m = 0;
            for (int x = 0; x < 8; ++x) m = std::max(m, part_h[8 + x]);
            if (m > 0) { part_max[0] = std::min(m, part_bound()); part_seen = part_max[0]; }
        }
    }

    void fill(sphmi_progress* out, int64_t steps) {
        if (!out) return;
        out->iteration = iteration; out->steps_done = steps; out->n_rebuilds = n_rebuilds;
        out->index_counter = index_counter; out->total_time = total_time; out->last_dt = last_dt; out->delta_x = delta_x;
    }

    // Queue one step of the while loop at src/SPHCellList.jl:742-802 with every per-step decision on the device
    // (k_step_control): Δx, Δt, the loop bound and the rebuild criterion.  Kernels of a cancelled step return at once.
    void serve_reschedules() {
        if (resched0_pending) { resched0_pending = false; reschedule_from_work(0); }
        if (resched1_pending) { resched1_pending = false; reschedule_from_work(1); }
    }
    void enqueue_step() {
        serve_reschedules();
        const bool fused = batch_fused;
        if (!fused) {
            Ev ev = begin_phase(PH_TIMESTEP);
            hipLaunchKernelGGL(k_step_control<T>, dim3(1), dim3(1), 0, stream, red_cur(), ctrl_cur(), cfg.h, cfg.c0, cfg.CFL);
            end_phase(ev);
        }
        const bool fused_in_motion = fused && motions.n > 0;
        if (fused_in_motion) { progress_motion(0.0, nullptr, true); cpar ^= 1; rpar ^= 1; }      // :765, and the decisions of the step with it
        else progress_motion(0.0, ctrl_cur());                                 // :765
        const bool fused_in_mdbc = fused && cfg.mdbc == SPHMI_MDBC_SIMPLE;
        if (fused_in_mdbc) { run_mdbc(nullptr, true); cpar ^= 1; rpar ^= 1; }   // :772, and the decisions of the step with it
        else if (cfg.mdbc == SPHMI_MDBC_SIMPLE) run_mdbc(ctrl_cur());           // :772
        ForceParams<T> P1 = force_params(iA, iA, iH, 0.0);
        if (fused_in_motion) P1.ctrl = ctrl_cur();                              // decided by k_progress_motion, which also zeroed the corrector's slots
        else if (fused_in_mdbc) {
            // decided by k_mdbc: the predictor reads the block it wrote and zeroes what that kernel could not
            P1.ctrl = ctrl_cur();
            P1.mdbc_zero = red_cur(); P1.mdbc_flag_zero = red_d + 4 * (rpar ^ 1) + 3;
        } else if (fused) {
            // the predictor takes the decisions: reads control block / slots `cpar` / `rpar`, leaves the other ones to the corrector
            P1.ctl_in = ctrl_d + cpar; P1.ctl_out = ctrl_d + (cpar ^ 1);
            P1.red_in = red_d + 4 * rpar; P1.red_zero = red_d + 4 * (rpar ^ 1);
            P1.ctl_h = cfg.h; P1.ctl_c0 = cfg.c0; P1.ctl_CFL = cfg.CFL;
            cpar ^= 1; rpar ^= 1;
        } else P1.ctrl = ctrl_cur();
        Ev e1 = begin_phase(PH_PASS1);
        launch_force<PASS_PREDICTOR>(P1);                                      // :774-781
        end_phase(e1);
        progress_motion(0.0, ctrl_cur());                                      // :787
        ForceParams<T> P2 = force_params(iH, iA, iB, 0.0); P2.ctrl = ctrl_cur();      // (force_params: P2.red = the current slots)
        Ev e2 = begin_phase(PH_PASS2);
        launch_force<PASS_CORRECTOR>(P2);                                      // :789-798
        end_phase(e2);
        std::swap(iA, iB);
    }

    void advance(double t_target, int64_t max_steps, sphmi_progress* out) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_advance before sphmi_upload");
        HC(hipSetDevice(cfg.device));
        delta_x = 1.0 + cfg.h;                                                // :739
        int64_t steps = 0;
        StepCtrl c{};
        c.delta_x = delta_x; c.total_time = total_time; c.t_step_start = total_time; c.t_target = t_target;
        c.max_steps = max_steps; c.last_dt = last_dt;
        try {
            // The first iteration of the loop always rebuilds (Δx = 1 + h ≥ h, :739,758): when there will BE a first iteration the
            // rebuild is run now and the control is told (StepCtrl::pre_rebuilt) — not a one-step batch that the control cancels.
            const bool pre = total_time <= t_target && max_steps != 0;
            if (pre) { c.pre_rebuilt = 1; delta_x = 0.0; rebuild_any(c, 1); }
            else { *ctrl_h = c; HC(hipMemcpyAsync(ctrl_cur(), ctrl_h, sizeof(StepCtrl), hipMemcpyHostToDevice, stream)); }
            bool fresh = pre;            // the first queued step follows a rebuild: it runs with Δx = 0 and adds nothing to it
            for (;;) {
                const int a0 = iA, b0 = iB;
                const int rpar0 = rpar;
                batch_fused = fused_control();
                const int64_t before = steps;
                const double dx0 = delta_x;
                // Queue up to the step that is expected to ask for the rebuild (Δx grows by 4·max|Δx| a step, slowly changing): the
                // n-th control that ADDS finds Δx = dx0 + n·rate, so the asking one is n* = ⌈(h − dx0) / rate⌉ — one later when the
                // batch opens behind a rebuild, whose step adds nothing (round 3 stopped one short there: every rebuild was asked for
                // by the lone first step of the next batch, a second synchronisation).  One spare step on top: a step queued behind
                // the asking one is cancelled and costs a few empty launches, a batch that ends before it a whole round trip.
                int batch = kBatch;
                if (dx0 >= cfg.h) batch = 1;
                else if (dx_rate > 0.0) batch = (int)std::max(1.0, std::min((double)kBatch, std::ceil((cfg.h - dx0) / dx_rate) + (fresh ? 2.0 : 1.0)));
                if (max_steps >= 0) batch = (int)std::min<int64_t>(batch, std::max<int64_t>(max_steps - steps, 1));
                for (int k = 0; k < batch; ++k) { batch_step = k; enqueue_step(); iteration += 1; }      // iteration: provisional (event sampling)
                batch_step = -1;
                // the block the last queued step wrote, both sets of slots (below: the bad-ρ flag), the counters and the run table of a
                // device-side rebuild: one copy
                HC(hipMemcpyAsync(ctl_m, ctl_d, kCtlBytes, hipMemcpyDeviceToHost, stream));
                // (the XCD finishing times of a sampled launch: the host-side rebuild reads them at its own synchronisation; handles that
                // rebuild on the device have none — the shares of the next measured-work schedule would never move: 3 % on the 159 k-particle
                // LaminarSPS case)
                const bool clocks = xcd_sampled;
                if (clocks) HC(hipMemcpyAsync(xcd_clock_h, xcd_clock_d, 16 * 8, hipMemcpyDeviceToHost, stream));
                HC(hipStreamSynchronize(stream));
                if (clocks) apply_xcd_feedback();
                *ctrl_h = ((const StepCtrl*)ctl_m)[cpar];
                sync_and_collect(ctrl_h, before, /*mirror_fresh=*/true);
                c = *ctrl_h;
                steps = c.steps_done;
                const int64_t executed = steps - before;
                // (control inside the predictor: the slots flipped at queue time, once per queued step; what counts is where
                // the last EXECUTED corrector left its maxima — cancelled steps consume nothing and zero nothing)
                if (batch_fused) rpar = rpar0 ^ (int)(executed & 1);
                iteration += executed - batch;                                    // what really ran
                // the state sets rotate once per EXECUTED step
                iA = (executed & 1) ? b0 : a0; iB = (executed & 1) ? a0 : b0;
                if (executed > 0) stepped = true;
                total_time = c.total_time; last_dt = c.last_dt; delta_x = c.delta_x;
                {
                    const int64_t grown = executed + (c.need_rebuild ? 1 : 0) - (fresh && executed > 0 ? 1 : 0);    // controls that added their 4·max|Δx|
                    if (grown > 0 && dx0 < cfg.h && c.delta_x > dx0) dx_rate = (c.delta_x - dx0) / (double)grown;
                }
                if (executed > 0) fresh = false;
                if (c.error == 2) throw EngineError(SPHMI_ERR_NUMERIC, "non-positive density produced (sign of ρ carries the MotionLimiter flag)");
                if (c.error == 3) {
                    // a particle left the grid of the last device-side rebuild: that rebuild copied instead of permuting and the
                    // control cancelled every step behind it — the same rebuild again, on the host, with a new grid
                    n_rebuilds -= 1; n_device_rebuilds -= 1; n_grid_overflows += 1;      // (one UpdateNeighbors! call of the reference)
                    rebuild();
                    c.error = 0;
                    *ctrl_h = c;
                    HC(hipMemcpyAsync(ctrl_cur(), ctrl_h, sizeof(StepCtrl), hipMemcpyHostToDevice, stream));
                    continue;
                }
                if (c.error) {
                    char buf[160];
                    snprintf(buf, sizeof(buf), "non-positive or NaN dt (%g) at iteration %lld (visc %g, |a|max %g, Δx %g)",
                             c.dt, (long long)iteration, c.last_visc, c.last_amax, c.delta_x);
                    throw EngineError(SPHMI_ERR_NUMERIC, buf);
                }
                if (c.need_rebuild) {                                             // :758-762
                    c.delta_x = 0.0; c.need_rebuild = 0; delta_x = 0.0;           // resume stays set: the queued step re-uses its Δt
                    rebuild_any(c, 2);
                    fresh = true;
                    continue;
                }
                if (c.stop || !(total_time <= t_target) || (max_steps >= 0 && steps >= max_steps)) {
                    // The corrector of the LAST executed step may have produced a non-positive density: no control runs after it in
                    // this call, so its flag (slot 3 of the set that corrector filled) is looked at here — the state must not be
                    // handed out as if it were good (tests/test_fuzz_gpu.py).
                    if (red_h[4 * rpar + 3] != 0)
                        throw EngineError(SPHMI_ERR_NUMERIC, "non-positive density produced (sign of ρ carries the MotionLimiter flag)");
                    break;
                }
            }
        } catch (...) { fill(out, steps); throw; }
        fill(out, steps);
    }

    // ---- upload / download --------------------------------------------------------------------
    template <class H> void pack_host(const void* position, const void* velocity, const void* acceleration,
                                      const void* density, const uint8_t* ty, const void* ghost_points,
                                      std::vector<V4>& h0, std::vector<V4>& h1, std::vector<V4>& ha, std::vector<V4>& hg, std::vector<V4>& hc) {
        const H* x = (const H*)position; const H* v = (const H*)velocity; const H* a = (const H*)acceleration;
        const H* r = (const H*)density; const H* g = (const H*)ghost_points;
        for (int i = 0; i < N; ++i) {
            V4 p0{}, p1{}, pa{}, pg{};
            p0.x = (T)x[i * D]; p0.y = (T)x[i * D + 1]; p0.z = D == 3 ? (T)x[i * D + 2] : T(0);
            p0.w = ty[i] == SPHMI_FLUID ? (T)r[i] : -(T)r[i];
            p1.x = (T)v[i * D]; p1.y = (T)v[i * D + 1]; p1.z = D == 3 ? (T)v[i * D + 2] : T(0); p1.w = 0;
            if (a) { pa.x = (T)a[i * D]; pa.y = (T)a[i * D + 1]; pa.z = D == 3 ? (T)a[i * D + 2] : T(0); }
            if (g) {
                pg.x = (T)g[i * D]; pg.y = (T)g[i * D + 1]; pg.z = D == 3 ? (T)g[i * D + 2] : T(0);
                bool nz = false;
                for (int d = 0; d < D; ++d) nz |= g[i * D + d] != H(0);
                pg.w = nz ? T(1) : T(0);
            }
            h0[i] = p0; h1[i] = p1; ha[i] = pa; hg[i] = pg;
            if (!hc.empty()) {
                // what the caller's value holds beyond the fp32 record (zero for a Float32 caller)
                V4 c{};
                c.x = (T)((double)x[i * D] - (double)p0.x); c.y = (T)((double)x[i * D + 1] - (double)p0.y);
                c.z = D == 3 ? (T)((double)x[i * D + 2] - (double)p0.z) : T(0);
                c.w = (T)((double)r[i] - (double)(T)r[i]);
                hc[i] = c;
            }
        }
    }

    void upload(const void* position, const void* velocity, const void* acceleration, const void* density,
                const uint8_t* ty, const int64_t* ids, const uint64_t* groups, const void* ghost_points) override {
        if (!position || !velocity || !density || !ty || !ids) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: null array");
        HC(hipSetDevice(cfg.device));
        for (int i = 0; i < N; ++i) {
            if (ty[i] < 1 || ty[i] > 3) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: ParticleType must be 1, 2 or 3");
        }
        std::vector<V4> h0(N), h1(N), ha(N), hg(N), hc(comp[0] ? N : 0);
        if (cfg.host_float_bytes == 8) pack_host<double>(position, velocity, acceleration, density, ty, ghost_points, h0, h1, ha, hg, hc);
        else pack_host<float>(position, velocity, acceleration, density, ty, ghost_points, h0, h1, ha, hg, hc);
        for (int i = 0; i < N; ++i)
            if (!(std::fabs((double)h0[i].w) > 0.0)) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: density must be positive");
        iA = 0; iH = 1; iB = 2; cur = 0;
        ghost_given = ghost_points != nullptr;
        {
            int ng = 0;
            for (int i = 0; i < N; ++i) ng += hg[i].w != T(0) ? 1 : 0;
            if (ng != mdbc_n_list) { (void)hipFree(mdbc_list_d); mdbc_list_d = nullptr; }
            mdbc_n_list = ng; mdbc_list_valid = false;
        }
        const size_t n = (size_t)N;
        std::vector<V4> hrec(2 * n);                                     // the two packets of a particle side by side
        for (size_t i = 0; i < n; ++i) { hrec[2 * i] = h0[i]; hrec[2 * i + 1] = h1[i]; }
        bounce.h2d(rec[iA], hrec.data(), 2 * n * sizeof(V4), stream);
        bounce.h2d(acc[cur], ha.data(), n * sizeof(V4), stream);
        bounce.h2d(ghost[cur], hg.data(), n * sizeof(V4), stream);
        if (comp[cur]) bounce.h2d(comp[cur], hc.data(), n * sizeof(V4), stream);
        bounce.h2d(type[cur], ty, n, stream);
        bounce.h2d(id[cur], ids, n * 8, stream);
        if (groups) bounce.h2d(grp[cur], groups, n * 8, stream);
        else HC(hipMemsetAsync(grp[cur], 0, n * 8, stream));
        HC(hipMemsetAsync(key[cur], 0, n * 4, stream));
        HC(hipMemsetAsync(red_d, 0, 8 * 8, stream));
        cpar = 0; rpar = 0;
        const int nb256 = (N + 255) / 256;
        hipLaunchKernelGGL(k_iota, dim3(nb256), dim3(256), 0, stream, prow[cur], N);
        // Pressure! (src/SPHCellList.jl:835) and the reductions Δt / update_delta_x! will read first
        hipLaunchKernelGGL(k_eos<T>, dim3(nb256), dim3(256), 0, stream, pk0[iA], pk1[iA], N, (T)cfg.rho0,
                           (T)(1.0 / cfg.rho0), (T)((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0));
        hipLaunchKernelGGL(k_init_reduce<T>, dim3(nb256), dim3(256), 0, stream, pk0[iA], pk1[iA], acc[cur], N,
                           (T)cfg.h, (T)cfg.eta2, red_d);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(stream));
        uploaded = true; stepped = false; have_grid = false; index_counter = 0;
    }

    // ---- the bench lattice generated on the device (no host arrays, no upload) ---------------------
    static int rnd(double x) { return (int)std::floor(x + 0.5); }
    static DamBreakGrid dam_break_grid(double dp, const sphmi_config& c) {
        DamBreakGrid G{};
        G.nx = rnd(1.6 / dp) + 1; G.ny = rnd(0.66 / dp) + 1;
        G.kwall = rnd(0.40 / dp); G.kcap = rnd(0.44 / dp); G.nk = std::max(G.kwall, G.kcap) + 1;
        G.pi0 = rnd(0.90 / dp); G.pi1 = G.pi0 + rnd(0.12 / dp); G.pj0 = rnd(0.22 / dp); G.pj1 = G.pj0 + rnd(0.14 / dp);
        G.fi = rnd(0.38 / dp) + 1; G.fj = rnd(0.62 / dp) + 1; G.fk = rnd(0.28 / dp) + 1;
        G.dp = dp; G.rho0 = c.rho0; G.g = c.g; G.B = c.c0 * c.c0 * c.rho0 / 7.0;
        return G;
    }
    void generate_dam_break_3d(double dp) override {
        if (D != 3) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_generate_dam_break_3d: the handle is not 3-D");
        if (!(dp > 0)) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_generate_dam_break_3d: dp must be positive");
        HC(hipSetDevice(cfg.device));
        const DamBreakGrid G = dam_break_grid(dp, cfg);
        const long long M = (long long)G.nx * G.ny * G.nk;
        const long long nf = (long long)G.fi * G.fj * G.fk;
        if (M > (1ll << 31) - 4096) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_generate_dam_break_3d: lattice too fine for one device");
        int *flag = nullptr, *pos = nullptr, *tsum = nullptr, *tot_d = nullptr;
        const int ntiles = (int)((M + kScanTile - 1) / kScanTile);
        HC(hipMalloc(&flag, (size_t)M * 4)); HC(hipMalloc(&pos, (size_t)M * 4)); HC(hipMalloc(&tsum, (size_t)(ntiles + 2) * 4)); HC(hipMalloc(&tot_d, 16));
        auto release = [&]() { (void)hipFree(flag); (void)hipFree(pos); (void)hipFree(tsum); (void)hipFree(tot_d); };
        try {
            iA = 0; iH = 1; iB = 2; cur = 0; ghost_given = false; mdbc_n_list = 0; mdbc_list_valid = false;
            int base = 0;
            const unsigned nbM = (unsigned)((M + 255) / 256);
            for (int which = 1; which <= 2; ++which) {              // the tank first, then the pillar object
                HC(hipMemsetAsync(tot_d, 0, 16, stream));
                hipLaunchKernelGGL(k_gen_flags, dim3(nbM), dim3(256), 0, stream, G, M, which, flag);
                hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(kScanThreads), 0, stream, (const int*)flag, pos, (int)M, tsum, tot_d);
                hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tsum, ntiles, tot_d + 1);
                hipLaunchKernelGGL(k_scan_add_nototal, dim3(ntiles), dim3(kScanThreads), 0, stream, pos, (int)M, (const int*)tsum);
                int tot[2] = {0, 0};
                bounce.d2h(tot, tot_d, 8, stream);
                if ((long long)base + tot[1] + nf > cap) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_generate_dam_break_3d: n_particles of the handle is smaller than the lattice (sphmi_dam_break_3d_count)");
                hipLaunchKernelGGL(k_gen_boundary<T>, dim3(nbM), dim3(256), 0, stream, G, M, (const int*)flag, (const int*)pos, base, pk0[iA], pk1[iA],
                                   type[cur], id[cur], grp[cur], comp[cur]);
                HC(hipGetLastError());
                base += tot[1];
            }
            if ((long long)base + nf != (long long)cap) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_generate_dam_break_3d: n_particles of the handle differs from the lattice (sphmi_dam_break_3d_count)");
            N = cap;
            hipLaunchKernelGGL(k_gen_fluid<T>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, stream, G, base, (int)nf, pk0[iA], pk1[iA], type[cur], id[cur], grp[cur], comp[cur]);
            const size_t n = (size_t)N;
            HC(hipMemsetAsync(acc[cur], 0, n * sizeof(V4), stream)); HC(hipMemsetAsync(ghost[cur], 0, n * sizeof(V4), stream));
            HC(hipMemsetAsync(key[cur], 0, n * 4, stream)); HC(hipMemsetAsync(red_d, 0, 8 * 8, stream)); cpar = 0; rpar = 0;
            const int nb256 = (N + 255) / 256;
            hipLaunchKernelGGL(k_iota, dim3(nb256), dim3(256), 0, stream, prow[cur], N);
            hipLaunchKernelGGL(k_eos<T>, dim3(nb256), dim3(256), 0, stream, pk0[iA], pk1[iA], N, (T)cfg.rho0, (T)(1.0 / cfg.rho0), (T)((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0));
            hipLaunchKernelGGL(k_init_reduce<T>, dim3(nb256), dim3(256), 0, stream, pk0[iA], pk1[iA], acc[cur], N, (T)cfg.h, (T)cfg.eta2, red_d);
            HC(hipGetLastError());
            HC(hipStreamSynchronize(stream));
        } catch (...) { release(); throw; }
        release();
        uploaded = true; stepped = false; have_grid = false; index_counter = 0;
    }

    template <class H> static void unpack3(const std::vector<V4>& s, H* out, int N, int D) {
        for (int i = 0; i < N; ++i) {
            out[i * D] = (H)s[i].x; out[i * D + 1] = (H)s[i].y;
            if (D == 3) out[i * D + 2] = (H)s[i].z;
        }
    }

    // ---- output side: device-packed fields, one copy per field ----------------------------------
    char* out_arena = nullptr; size_t out_arena_bytes = 0;
    std::vector<std::pair<void*, size_t>> host_pinned;
    HostBounce bounce;          // every copy to or from pageable host memory goes through it
    // Page-locking is the CALLER's decision (sphmi_host_register): only the caller knows that an array outlives the
    // handle's use of it — a registration that survived a free + reuse of the address range would be a stale mapping.
    void host_register(void* p, size_t bytes) override {
        if (!p || !bytes) return;
        for (auto& e : host_pinned) if (e.first == p) return;
        HC(hipSetDevice(cfg.device));
        if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) host_pinned.emplace_back(p, bytes);
        else { (void)hipGetLastError(); throw EngineError(SPHMI_ERR_DEVICE, "sphmi_host_register: hipHostRegister failed"); }
    }
    void host_unregister(void* p) override {
        for (size_t k = 0; k < host_pinned.size(); ++k) if (host_pinned[k].first == p) {
            (void)hipHostUnregister(p);
            host_pinned.erase(host_pinned.begin() + (long)k);
            return;
        }
    }
    hipStream_t copy_stream = nullptr; hipEvent_t ev_packed = nullptr; bool download_pending = false;
    bool dl_dst_page_locked = false;               // slab engines: the multi-device handle downloads into its own page-locked staging
    unsigned long long* dl_tags_host = nullptr;    // slab engines: the order tags travel with a download (the multi-device handle merges by them)
    int out_comp = 0;          // components per output vector: D, or 3 (sphmi_set_output_components)
    void set_output_components(int c) override {
        if (c != D && c != 3) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_output_components: dims or 3");
        out_comp = c;
    }
    // begin: snapshot every requested field into the arena IN STREAM ORDER (so later steps cannot disturb it), then
    // hand the device→host copies to a second stream; end: wait for them.  Between the two the caller may advance.
    template <class H>
    void download_begin_as(void* position, void* velocity, void* acceleration, void* density, void* pressure,
                           int64_t* ids, uint8_t* ty, uint64_t* groups, void* ghost_points, int64_t* cells) {
        if (!copy_stream) { HC(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); HC(hipEventCreateWithFlags(&ev_packed, hipEventDisableTiming)); }
        if (download_pending) download_end();
        const size_t n = (size_t)N, nd = n * (size_t)out_comp, ncell_d = n * (size_t)D;
        const size_t need = (4 * nd + 2 * n) * sizeof(H) + ncell_d * 8 + n * 25 + 256 * 13;
        if (need > out_arena_bytes) {
            (void)hipFree(out_arena);
            out_arena = nullptr; out_arena_bytes = 0;
            HC(hipMalloc(&out_arena, need));
            out_arena_bytes = need;
        }
        char* cursor = out_arena;
        auto take = [&](bool wanted, size_t bytes) -> char* {
            if (!wanted) return nullptr;
            char* r = cursor; cursor += (bytes + 255) & ~size_t(255); return r;
        };
        OutFields<H> o{};
        o.pos = (H*)take(position, nd * sizeof(H)); o.vel = (H*)take(velocity, nd * sizeof(H));
        o.acc = (H*)take(acceleration, nd * sizeof(H)); o.rho = (H*)take(density, n * sizeof(H));
        o.press = (H*)take(pressure, n * sizeof(H)); o.ghost = (H*)take(ghost_points, nd * sizeof(H));
        o.cells = (long long*)take(cells, ncell_d * 8);
        char* a_id = take(ids, n * 8); char* a_ty = take(ty, n); char* a_grp = take(groups, n * 8);
        char* a_tag = take(dl_tags_host != nullptr && otag[0], n * 8);
        hipLaunchKernelGGL((k_pack_output<T, H>), dim3((N + 255) / 256), dim3(256), 0, stream, pk0[iA], pk1[iA],
                           stepped ? Half<const V4>(pk0[iH]) : Half<const V4>(), acc[cur], ghost[cur], comp[cur], key[cur], N, D, out_comp, grid, have_grid ? 1 : 0,
                           (T)cfg.rho0, (T)(1.0 / cfg.rho0), (T)((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0), o);
        HC(hipGetLastError());
        if (a_id) HC(hipMemcpyAsync(a_id, id[cur], n * 8, hipMemcpyDeviceToDevice, stream));
        if (a_ty) HC(hipMemcpyAsync(a_ty, type[cur], n, hipMemcpyDeviceToDevice, stream));
        if (a_grp) HC(hipMemcpyAsync(a_grp, grp[cur], n * 8, hipMemcpyDeviceToDevice, stream));
        if (a_tag) HC(hipMemcpyAsync(a_tag, otag[cur], n * 8, hipMemcpyDeviceToDevice, stream));
        HC(hipEventRecord(ev_packed, stream));
        HC(hipStreamWaitEvent(copy_stream, ev_packed, 0));
        // arrays the caller page-locked (sphmi_host_register) are written by the copy engine while the caller advances; any
        // other array is filled from the snapshot in download_end, through the bounce buffer (HostBounce, above)
        out_deferred.clear();
        auto copy = [&](void* dst, const void* src, size_t bytes) {
            if (!dst || !bytes) return;
            if (dl_dst_page_locked || is_registered(dst, bytes)) HC(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, copy_stream));
            else out_deferred.push_back({dst, src, bytes});
        };
        copy(position, o.pos, nd * sizeof(H)); copy(velocity, o.vel, nd * sizeof(H));
        copy(acceleration, o.acc, nd * sizeof(H)); copy(density, o.rho, n * sizeof(H));
        copy(pressure, o.press, n * sizeof(H)); copy(ghost_points, o.ghost, nd * sizeof(H));
        copy(cells, o.cells, ncell_d * 8);
        copy(ids, a_id, n * 8); copy(ty, a_ty, n); copy(groups, a_grp, n * 8);
        if (a_tag) copy(dl_tags_host, a_tag, n * 8);
        download_pending = true;
    }
    struct Deferred { void* dst; const void* src; size_t bytes; };
    std::vector<Deferred> out_deferred;
    bool is_registered(const void* q, size_t bytes) const {
        for (auto& e : host_pinned) if ((const char*)q >= (const char*)e.first && (const char*)q + bytes <= (const char*)e.first + e.second) return true;
        return false;
    }
    void download_begin(void* position, void* velocity, void* acceleration, void* density, void* pressure,
                        int64_t* ids, uint8_t* ty, uint64_t* groups, void* ghost_points, int64_t* cells) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_download before sphmi_upload");
        HC(hipSetDevice(cfg.device));
        if (cfg.host_float_bytes == 8) download_begin_as<double>(position, velocity, acceleration, density, pressure, ids, ty, groups, ghost_points, cells);
        else download_begin_as<float>(position, velocity, acceleration, density, pressure, ids, ty, groups, ghost_points, cells);
    }
    void download_end() override {
        if (!download_pending) return;
        HC(hipSetDevice(cfg.device));
        HC(hipStreamSynchronize(copy_stream));
        download_pending = false;
        for (const Deferred& d : out_deferred) bounce.d2h(d.dst, d.src, d.bytes, copy_stream);
        out_deferred.clear();
    }
    void download(void* position, void* velocity, void* acceleration, void* density, void* pressure,
                  int64_t* ids, uint8_t* ty, uint64_t* groups, void* ghost_points, int64_t* cells) override {
        download_begin(position, velocity, acceleration, density, pressure, ids, ty, groups, ghost_points, cells);
        download_end();
    }

    void download_kernel_output(void* kernel, void* kernel_gradient) override {
        if (!kout_d) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_kernel_output: the handle was not created with kernel_output = STORE");
        HC(hipSetDevice(cfg.device));
        HC(hipStreamSynchronize(stream));
        std::vector<V4> tmp(N);
        bounce.d2h(tmp.data(), kout_d, (size_t)N * sizeof(V4), stream);
        const bool h8 = cfg.host_float_bytes == 8;
        if (kernel_gradient) { if (h8) unpack3(tmp, (double*)kernel_gradient, N, D); else unpack3(tmp, (float*)kernel_gradient, N, D); }
        if (kernel) for (int i = 0; i < N; ++i) { if (h8) ((double*)kernel)[i] = (double)tmp[i].w; else ((float*)kernel)[i] = (float)tmp[i].w; }
    }
    // sphmi_download_permutation: the sort carried every particle's row at the previous call along (prow); hand it over and
    // start counting from the present order.  The reference's sort! permutes all 17 fields of the StructArray
    // (src/SPHCellList.jl:142); the engine carries ten of them, and this is what lets the caller permute the rest.
    void download_permutation(int64_t* prev_row) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_permutation before sphmi_upload");
        if (!prev_row) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_download_permutation: null array");
        HC(hipSetDevice(cfg.device));
        std::vector<int> tmp((size_t)N);
        bounce.d2h(tmp.data(), prow[cur], (size_t)N * 4, stream);
        hipLaunchKernelGGL(k_iota, dim3((N + 255) / 256), dim3(256), 0, stream, prow[cur], N);
        HC(hipGetLastError());
        HC(hipStreamSynchronize(stream));
        for (int i = 0; i < N; ++i) prev_row[i] = tmp[(size_t)i];
    }
    // Pressure! + [mDBC] + ONE forces-only neighbour pass on the current cell list; {a, dρ/dt} of every particle held are left
    // in the scratch record array rec[iB] (N contiguous packets), SimParticles.Acceleration survives.  all_lists: a slab
    // engine runs its interior and its slab-edge tiles (the ghost layers must be current: the caller has just rebuilt).
    void forces_local(int apply_mdbc, bool all_lists) {
        const int nb256 = (N + 255) / 256;
        hipLaunchKernelGGL(k_eos<T>, dim3(nb256), dim3(256), 0, stream, pk0[iA], pk1[iA], N, (T)cfg.rho0,
                           (T)(1.0 / cfg.rho0), (T)((cfg.c0 * cfg.c0 * cfg.rho0) / 7.0));
        if (apply_mdbc && cfg.mdbc == SPHMI_MDBC_SIMPLE) run_mdbc();
        ForceParams<T> P = force_params(iA, iA, iH, 0.0);
        P.accbuf = rec[iB];                       // (scratch: N contiguous packets at the start of the third record array)
        Ev e1 = begin_phase(PH_PASS1);
        launch_force<PASS_FORCES_ONLY>(P);
        end_phase(e1);
        if (all_lists && part_max[1] > 0) {
            Ev e2 = begin_phase(PH_PASS1_EDGE);
            launch_force<PASS_FORCES_ONLY>(P, 1);
            end_phase(e2);
        }
    }
    void forces_once(int apply_mdbc, void* drhodt, void* acceleration) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_forces_once before sphmi_upload");
        HC(hipSetDevice(cfg.device));
        rebuild();
        forces_local(apply_mdbc, false);
        sync_and_collect();
        std::vector<V4> tmp(N);
        bounce.d2h(tmp.data(), rec[iB], (size_t)N * sizeof(V4), stream);
        const bool h8 = cfg.host_float_bytes == 8;
        if (acceleration) { if (h8) unpack3(tmp, (double*)acceleration, N, D); else unpack3(tmp, (float*)acceleration, N, D); }
        if (drhodt) for (int i = 0; i < N; ++i) { if (h8) ((double*)drhodt)[i] = (double)tmp[i].w; else ((float*)drhodt)[i] = (float)tmp[i].w; }
    }

    // UniqueCells[2:IndexCounter] (src/SPHCellList.jl:148-157) for the grid export of the output side: heads of the key
    // runs, compacted on the device (tmp_idx / perm are scratch between rebuilds); the host receives the cell coordinates
    void unique_cells(int64_t* out, int64_t cap, int64_t* n_out) override {
        HC(hipSetDevice(cfg.device));
        if (!have_grid) { if (n_out) *n_out = 0; return; }
        const int nb256 = (N + 255) / 256;
        int *flag = tmp_idx, *pos = perm;              // N ints each; pos needs N + 1: the total goes through misc_d
        hipLaunchKernelGGL(k_cell_heads, dim3(nb256), dim3(256), 0, stream, (const int*)key[cur], N, grid.ncell, flag);
        const int ntiles = (N + kScanTile - 1) / kScanTile;
        HC(hipMemsetAsync(misc_d + 4, 0, 3 * 4, stream));
        hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(kScanThreads), 0, stream, (const int*)flag, pos, N, tile_tsum_u(ntiles), misc_d + 4);
        hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, stream, tile_tsum_u(ntiles), ntiles, misc_d + 5);
        hipLaunchKernelGGL(k_scan_add_nototal, dim3(ntiles), dim3(kScanThreads), 0, stream, pos, N, (const int*)tile_tsum_u(ntiles));
        HC(hipGetLastError());
        HC(hipMemcpyAsync(misc_h + 4, misc_d + 4, 2 * 4, hipMemcpyDeviceToHost, stream));
        HC(hipStreamSynchronize(stream));
        const int64_t nu = misc_h[5];
        if (n_out) *n_out = nu;
        if (!out || nu == 0) return;
        if (cap < nu) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_unique_cells: capacity too small");
        long long* od = nullptr;
        HC(hipMalloc(&od, (size_t)nu * D * 8));
        hipLaunchKernelGGL(k_cells_out, dim3(nb256), dim3(256), 0, stream, (const int*)key[cur], (const int*)flag, (const int*)pos, N, grid, D, od);
        hipError_t e1 = hipGetLastError();
        try { HC(e1); bounce.d2h(out, od, (size_t)nu * D * 8, stream); } catch (...) { (void)hipStreamSynchronize(stream); (void)hipFree(od); throw; }
        (void)hipFree(od);
    }
    int* uc_tsum = nullptr; int uc_tsum_n = 0;
    int* tile_tsum_u(int ntiles) {
        if (ntiles + 2 > uc_tsum_n) { (void)hipFree(uc_tsum); uc_tsum = nullptr; uc_tsum_n = ntiles + 64; HC(hipMalloc(&uc_tsum, (size_t)uc_tsum_n * 4)); }
        return uc_tsum;
    }


    // ---- domain decomposition: what the slab driver (sphmi_multi.h, MultiEngine) asks of one slab engine ---------------
    void dd_upload(int64_t n, const void* position, const void* velocity, const void* acceleration,
                   const void* density, const uint8_t* ty, const int64_t* ids, const uint64_t* groups,
                   const void* ghost_points, const int64_t* upload_index) {
        if (n < 1 || n > cap) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_dd_upload: particle count exceeds the handle's capacity");
        if (cfg.mdbc != SPHMI_MDBC_NONE && !ghost_points)
            throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_dd_upload: mDBC handle without ghost points");
        N = (int)n;
        upload(position, velocity, acceleration, density, ty, ids, groups, ghost_points);
        // order tags: before the first sort a particle's "previous sorted index" is its index in the caller's arrays
        for (int k = 0; k < 2; ++k) if (!otag[k]) HC(hipMalloc(&otag[k], (size_t)cap * 8));
        std::vector<unsigned long long> t((size_t)n);
        for (int64_t i = 0; i < n; ++i) t[i] = upload_index ? (unsigned long long)upload_index[i] : (unsigned long long)i;
        bounce.h2d(otag[cur], t.data(), (size_t)n * 8, stream);
        // … and its row for sphmi_download_permutation is its row in the caller's arrays (the column travels with the sorts,
        // the migration records and the ghost-layer records like every other)
        if (upload_index) {
            std::vector<int> rows((size_t)n);
            for (int64_t i = 0; i < n; ++i) rows[(size_t)i] = (int)upload_index[i];
            bounce.h2d(prow[cur], rows.data(), (size_t)n * 4, stream);
        }
    }
    // ProgressMotion of the queued step (src/SPHCellList.jl:765,787) on owned particles AND ghost copies — a prescribed
    // motion is the same function of time on every rank — before the halo of the pass is packed
    void dd_progress_motion() {
        if (!dd_ctrl_on) throw EngineError(SPHMI_ERR_STATE, "sphmi_dd_progress_motion needs sphmi_dd_ctrl_init");
        HC(hipSetDevice(cfg.device));
        progress_motion(0.0, ctrl_d);
    }
    // mDBC (:772) for every boundary particle held, ghost copies included: with a halo wide enough (tests/slab_planner_reference.py holds the planner in numpy)
    // the copies a pass can see get the owner's value up to summation order, and nothing has to travel twice
    void dd_mdbc() {
        if (cfg.mdbc != SPHMI_MDBC_SIMPLE) return;
        HC(hipSetDevice(cfg.device));
        run_mdbc(dd_ctrl_on ? ctrl_d : nullptr);
    }
    void reset_count() override { N = cap; }
    // the same two arrays into DEVICE buffers of the caller, in stream order (no host round trip: the driver builds its
    // migration / ghost-layer / halo index lists with device-side compaction)
    void dd_cell_x_dev(int32_t* out_dev) {
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_cellx<T>, dim3((N + 255) / 256), dim3(256), 0, stream, pk0[iA], N, (T)cfg.H_inv, dd_axis, (int*)out_dev);
        HC(hipGetLastError());
    }
    void dd_column_cost(int64_t col0, int32_t ncols, uint64_t* out_dev) {
        if (!have_grid) throw EngineError(SPHMI_ERR_STATE, "sphmi_dd_column_cost before the first rebuild");
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_column_cost, dim3((N + 255) / 256), dim3(256), 0, stream, key[cur], type[cur], cstart, N, grid, D, dd_axis,
                           (long long)col0, (int)ncols, (unsigned long long*)out_dev);
        HC(hipGetLastError());
    }
    void dd_gather(const int32_t* idx_dev, int64_t n, void* buf_dev) {
        if (n <= 0) return;
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_gather<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pk0[iA], pk1[iA], acc[cur],
                           ghost[cur], comp[cur], id[cur], grp[cur], otag[cur], prow[cur], type[cur], idx_dev, (int)n, buf_dev);
        HC(hipGetLastError());
    }
    void dd_kill(const int32_t* idx_dev, int64_t n) {
        if (n <= 0) return;
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_kill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, type[cur], idx_dev, (int)n);
        HC(hipGetLastError());
    }
    void dd_kill_ghosts() {
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_kill_ghosts, dim3((N + 255) / 256), dim3(256), 0, stream, type[cur], N);
        HC(hipGetLastError());
    }
    void dd_append(const void* buf_dev, int64_t n, int flag) {
        if (n <= 0) return;
        if ((int64_t)N + n > cap) throw EngineError(SPHMI_ERR_DOMAIN, "domain decomposition: rank capacity exceeded (too many arrivals)");
        HC(hipSetDevice(cfg.device));
        hipLaunchKernelGGL(k_dd_append<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pk0[iA], pk1[iA], acc[cur],
                           ghost[cur], comp[cur], id[cur], grp[cur], otag[cur], prow[cur], type[cur], N, (int)n, const_cast<void*>(buf_dev), (uint8_t)flag);
        HC(hipGetLastError());
        N += (int)n;
    }
    void dd_rebuild() {
        HC(hipSetDevice(cfg.device));
        rebuild();
        int live = 0;
        bounce.d2h(&live, cstart + grid.ncell, 4, stream);
        sync_and_collect();
        if (live < N) index_counter -= 1;      // the graveyard is not a cell
        N = live;
    }
    // ---- device-side step control for the slab driver: same k_step_control as Engine::advance, fed with the
    // MAX-allreduced reduction slots, so the host looks at the flags once per batch of queued steps --------------
    bool dd_ctrl_on = false; int dd_a0 = 0, dd_b0 = 0; int64_t dd_steps_at_sync = 0;
    void dd_ctrl_init(double dx0, double t_target, int64_t max_steps, bool pre_rebuilt) {
        HC(hipSetDevice(cfg.device));
        StepCtrl c{};
        c.delta_x = dx0; c.pre_rebuilt = pre_rebuilt ? 1 : 0; c.total_time = total_time; c.t_step_start = total_time; c.t_target = t_target;
        c.max_steps = max_steps; c.last_dt = last_dt;
        *ctrl_h = c;
        HC(hipMemcpyAsync(ctrl_d, ctrl_h, sizeof(StepCtrl), hipMemcpyHostToDevice, stream));
        dd_ctrl_on = true; dd_a0 = iA; dd_b0 = iB; dd_steps_at_sync = 0; batch_step = -1;
    }
    // the slab driver has queued this step's control (k_dd_merge_control on ctrl_d): the coming corrector fills slot set `fill`
    void dd_control_queued(int fill) { batch_step += 1; rpar = fill; }
    void dd_ctrl_sync(sphmi_dd_control* out) {
        HC(hipSetDevice(cfg.device));
        HC(hipMemcpyAsync(ctrl_h, ctrl_d, sizeof(StepCtrl), hipMemcpyDeviceToHost, stream));
        sync_and_collect(ctrl_h, dd_steps_at_sync);
        batch_step = -1;
        const StepCtrl c = *ctrl_h;
        const int64_t executed = c.steps_done - dd_steps_at_sync;
        // the state sets rotate once per EXECUTED step (the host rotated them once per QUEUED step)
        iA = (executed & 1) ? dd_b0 : dd_a0; iB = (executed & 1) ? dd_a0 : dd_b0;
        dd_a0 = iA; dd_b0 = iB; dd_steps_at_sync = c.steps_done;
        iteration += executed; total_time = c.total_time; last_dt = c.last_dt;
        if (executed > 0) stepped = true;
        if (out) {
            out->steps_done = c.steps_done; out->total_time = c.total_time; out->last_dt = c.last_dt; out->delta_x = c.delta_x;
            out->need_rebuild = c.need_rebuild; out->stop = c.stop; out->error = c.error; out->reserved = 0;
        }
    }
    void dd_ctrl_resume() {
        HC(hipSetDevice(cfg.device));
        ctrl_h->delta_x = 0.0; ctrl_h->need_rebuild = 0;          // `resume` stays set: the queued step re-uses its Δt
        HC(hipMemcpyAsync(ctrl_d, ctrl_h, sizeof(StepCtrl), hipMemcpyHostToDevice, stream));
        dd_a0 = iA; dd_b0 = iB;                                  // the rebuild may have swapped the sets
    }
    void dd_set_slab(int axis, int64_t lo, int64_t hi, int has_lo, int has_hi) {
        if (axis < 0 || axis >= D) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_dd_set_slab: axis out of range");
        dd_slab = true; dd_axis = axis; dd_col_lo = lo; dd_col_hi = hi; dd_has_lo = has_lo != 0; dd_has_hi = has_hi != 0;
    }
    // part: 0 = the whole pass, 1 = interior tiles only, 2 = slab-edge tiles only (after the halo has landed)
    void dd_pass(int which, double dt, int part) {
        HC(hipSetDevice(cfg.device));
        if (which != 1 && which != 2) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_dd_pass: which must be 1 or 2");
        ForceParams<T> P = which == 1 ? force_params(iA, iA, iH, dt) : force_params(iH, iA, iB, dt);
        if (dd_ctrl_on) P.ctrl = ctrl_d;
        if (part == 0 || part == 1) {
            Ev e = begin_phase(which == 1 ? PH_PASS1 : PH_PASS2);
            if (which == 1) launch_force<PASS_PREDICTOR>(P, 0); else launch_force<PASS_CORRECTOR>(P, 0);
            end_phase(e);
        }
        if ((part == 0 || part == 2) && part_max[1] > 0) {
            Ev e = begin_phase(which == 1 ? PH_PASS1_EDGE : PH_PASS2_EDGE);
            edge_overlapped = part == 2;
            try { if (which == 1) launch_force<PASS_PREDICTOR>(P, 1); else launch_force<PASS_CORRECTOR>(P, 1); }
            catch (...) { edge_overlapped = false; throw; }
            edge_overlapped = false;
            end_phase(e);
        }
        if (which == 2 && part != 1) {
            std::swap(iA, iB);
            if (!dd_ctrl_on) { stepped = true; iteration += 1; last_dt = dt; total_time += dt; }   // else: dd_ctrl_sync
        }
    }
    void timers(int32_t cap, const char** names, double* secs, int64_t* calls, int32_t* n) override {
        if (n) *n = PH_COUNT + 3;
        for (int i = 0; i < PH_COUNT && i < cap; ++i) {
            if (names) names[i] = kPhaseNames[i];
            if (secs) secs[i] = ph_secs[i];
            if (calls) calls[i] = ph_calls[i];
        }
        if (PH_COUNT < cap) {      // (not a phase of the reference: how many of the rebuilds above found every particle in its cell and sorted nothing)
            if (names) names[PH_COUNT] = "02b UpdateNeighbors calls that were the identity (no sort)";
            if (secs) secs[PH_COUNT] = 0.0;
            if (calls) calls[PH_COUNT] = n_identity_rebuilds;
        }
        if (PH_COUNT + 1 < cap) {
            if (names) names[PH_COUNT + 1] = "02c UpdateNeighbors calls served on the device (no host round trip)";
            if (secs) secs[PH_COUNT + 1] = dev_rebuild_secs;
            if (calls) calls[PH_COUNT + 1] = n_device_rebuilds;
        }
        if (PH_COUNT + 2 < cap) {
            if (names) names[PH_COUNT + 2] = "02d device-side rebuilds repeated on the host (a particle left the sticky grid)";
            if (secs) secs[PH_COUNT + 2] = 0.0;
            if (calls) calls[PH_COUNT + 2] = n_grid_overflows;
        }
    }
    void force_stats(int reset, double* avg_ms, int64_t* launches) override {
        if (avg_ms) *avg_ms = force_launches ? force_ms / (double)force_launches : 0.0;
        if (launches) *launches = force_launches;
        if (reset) { force_ms = 0; force_launches = 0; ev_always_until = iteration + 2; }   // short windows still get samples
    }
    void device_ptrs(void** p0, void** p1, int64_t* n) override {
        if (p0) *p0 = pk0[iA].p;            // packet h of particle i at p[2·i] (the two pointers are one array, 1 packet apart)
        if (p1) *p1 = pk1[iA].p;
        if (n) *n = N;
    }
};

}  // namespace sphmi

#include "sphmi_multi.h"

// ================================================================================================
// C ABI
// ================================================================================================
struct sphmi_handle { sphmi::EngineBase* e; };

#define SPHMI_GUARD(h, body)                                                        \
    if (!(h) || !(h)->e) return SPHMI_ERR_ARGUMENT;                                 \
    try { body; return SPHMI_OK; }                                                  \
    catch (const sphmi::EngineError& x) { (h)->e->err = x.what(); return x.status; } \
    catch (const std::exception& x) { (h)->e->err = x.what(); return SPHMI_ERR_DEVICE; }

extern "C" {

static_assert(SPHMI_ABI_VERSION == 5, "update the text of sphmi_backend_info");
const char* sphmi_backend_info(void) {
    return "sphmi abi 5 | HIP gfx950 (CDNA4, wave64) | kernels: neighbor_force<fp32|fp64, 2D|3D>, "
           "counting-sort cell list, mDBC, moving bodies, shifting | multi-device handles: slabs over device copies / RCCL "
           "(the real RCCL has run with one rank only; the RCCL branch with 2-4 ranks through a checking double) | no CPU fallback";
}

const char* sphmi_last_error(const sphmi_handle* h) {
    if (!h || !h->e) return sphmi::g_create_error.c_str();
    return h->e->err.c_str();
}

static int check_config(const sphmi_config* cfg, int slabs, std::string& why);

static int create_any(const sphmi_config* cfg, int32_t rank, int32_t world, const void* unique_id, sphmi_handle** out) {
    using namespace sphmi;
    auto fail = [&](int st, const std::string& m) { g_create_error = m; return st; };
    if (!cfg || !out) return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: null argument");
    std::string why;
    if (cfg->struct_size != (int32_t)sizeof(sphmi_config) || cfg->abi_version != SPHMI_ABI_VERSION)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: struct_size / abi_version mismatch");
    // device_float_bytes = 0: the library chooses the arithmetic (sphmi_auto_device_float_bytes, include/sphmi.h)
    sphmi_config resolved = *cfg;
    if (resolved.device_float_bytes == 0) {
        resolved.device_float_bytes = sphmi_auto_device_float_bytes(cfg);
        // the policy never turns a handle the fp32 kernels can hold into an error: beyond the fp64 kernels' 2^26 particles per device it stays with fp32
        const int64_t slabs = rank >= 0 ? std::max(world, 1) : std::max(cfg->n_devices, 1);
        const int64_t per = (cfg->n_particles + slabs - 1) / slabs;
        if (resolved.device_float_bytes == 8 && per > (1ll << 26) - 1 && per <= (1ll << 27) - 1) resolved.device_float_bytes = 4;
    }
    cfg = &resolved;
    // slabs the particle set is spread over: sphmi_create_rank's world, or the device list of sphmi_create
    if (int st = check_config(cfg, rank >= 0 ? std::max(world, 1) : std::max(cfg->n_devices, 1), why)) return fail(st, why);
    try {
        sphmi_handle* h = new sphmi_handle{nullptr};
        try {
            const bool f32 = cfg->device_float_bytes == 4;
            if (rank >= 0) {
                if (world < 1 || rank >= world) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_create_rank: rank / world out of range");
                if (f32) h->e = new MultiEngine<float>(*cfg, world, rank, unique_id); else h->e = new MultiEngine<double>(*cfg, world, rank, unique_id);
            } else if (cfg->n_devices > 1) {
                if (f32) h->e = new MultiEngine<float>(*cfg, cfg->n_devices, -1, nullptr); else h->e = new MultiEngine<double>(*cfg, cfg->n_devices, -1, nullptr);
            } else {
                sphmi_config c = *cfg;
                if (c.n_devices == 1) c.device = c.devices[0];
                if (f32) h->e = new Engine<float>(c); else h->e = new Engine<double>(c);
            }
        } catch (...) { delete h; throw; }
        *out = h;
        return SPHMI_OK;
    } catch (const EngineError& x) { return fail(x.status, x.what()); }
    catch (const std::exception& x) { return fail(SPHMI_ERR_DEVICE, x.what()); }
}

// The precision policy of `device_float_bytes = 0` (include/sphmi.h): fp32 kernels where every term of the path is CONTINUOUS in the
// particle positions, fp64 kernels where the reference's algorithm has a discontinuity that fp32 state decides differently:
//  * a kernel cut off BEFORE it vanishes (H = k·h, k < 2: example/DucklingMDBC.jl 1.5, example/MovingSquare2d.jl sqrt 2) — a pair at
//    r ~ H switches a finite force on or off (src/SPHCellList.jl:275), lattices with spacing H/2 put thousands of pairs exactly on
//    r = H, and fp32 handles leave the 1e-5 of the fp64 path within tens of steps.  With H >= 2h Wendland C2 and the cubic spline vanish
//    with their gradients at the cut: a pair that fp32 rounding puts on the other side contributes (nearly) nothing either way;
//  * mDBC (src/SPHCellList.jl:598-622): a ghost node keeps its density when it has no fluid neighbour, takes rho := b1/A11 — a 0/0 — when
//    its only neighbour sits at r ~ H, and switches between the full solve and that fallback at |det A| = 1e-3; an fp32 trajectory
//    (x off by 1e-7) crosses those lines a step early or late and freezes a different density into single boundary particles (1e-4
//    relative on a handful of them in a streaming Dambreak2dMDBC; every fluid particle stays within 4e-6).
// Measured on every stock example: BASELINE.md section 7, profiles/r05_fp32_examples_parity.md; tests/test_example_precision_gpu.py.
int32_t sphmi_auto_device_float_bytes(const sphmi_config* cfg) {
    if (!cfg) return 0;
    const bool vanishes_at_cut = cfg->H >= 2.0 * cfg->h * (1.0 - 1e-12);
    return (vanishes_at_cut && cfg->mdbc == SPHMI_MDBC_NONE) ? 4 : 8;
}
int sphmi_device_float_bytes(const sphmi_handle* h, int32_t* out) {
    if (!h || !h->e || !out) return SPHMI_ERR_ARGUMENT;
    *out = h->e->cfg.device_float_bytes;
    return SPHMI_OK;
}
int sphmi_create(const sphmi_config* cfg, sphmi_handle** out) { return create_any(cfg, -1, 0, nullptr, out); }
int sphmi_create_rank(const sphmi_config* cfg, int32_t rank, int32_t world, const void* unique_id, sphmi_handle** out) {
    if (rank < 0) { sphmi::g_create_error = "sphmi_create_rank: negative rank"; return SPHMI_ERR_ARGUMENT; }
    return create_any(cfg, rank, world, unique_id, out);
}
int sphmi_rccl_unique_id(void* id_out) {
    using namespace sphmi;
    if (!id_out) return SPHMI_ERR_ARGUMENT;
    try {
        ncclUniqueId id;
        NCX("ncclGetUniqueId", 0, -1, nullptr, Rccl::get().GetUniqueId(&id));
        memcpy(id_out, &id, sizeof id);
        return SPHMI_OK;
    } catch (const EngineError& x) { g_create_error = x.what(); return x.status; }
    catch (const std::exception& x) { g_create_error = x.what(); return SPHMI_ERR_DEVICE; }
}
int sphmi_rccl_probe(void) {
    using namespace sphmi;
    try { (void)Rccl::get(); return SPHMI_OK; }      // dlopen + every symbol the slab driver calls; no bootstrap root, no thread, no socket
    catch (const EngineError& x) { g_create_error = x.what(); return x.status; }
    catch (const std::exception& x) { g_create_error = x.what(); return SPHMI_ERR_DEVICE; }
}
int sphmi_shm_selftest(const void* unique_id, int32_t rank, int32_t world, int64_t n_bytes) {
    using namespace sphmi;
    if (!unique_id || world < 1 || rank < 0 || rank >= world || n_bytes < 0) return SPHMI_ERR_ARGUMENT;
    try {
        ShmWorld w(unique_id, 128, rank, world);
        // reductions: a vector longer than one reduce slot, SUM and MAX
        const size_t n = ShmWorld::kReduceWords + 37;
        std::vector<int64_t> v(n), u(n);
        for (size_t k = 0; k < n; ++k) { v[k] = (int64_t)(k % 101) * (rank + 1); u[k] = (int64_t)((k * 7 + (size_t)rank * 13) % 1009); }
        w.allreduce(v.data(), n, ShmWorld::SUM);
        w.allreduce(u.data(), n, ShmWorld::MAX);
        for (size_t k = 0; k < n; ++k) {
            if (v[k] != (int64_t)(k % 101) * world * (world + 1) / 2) { g_create_error = "shm selftest: SUM mismatch"; return SPHMI_ERR_STATE; }
            int64_t m = 0; for (int r = 0; r < world; ++r) m = std::max<int64_t>(m, (int64_t)((k * 7 + (size_t)r * 13) % 1009));
            if (u[k] != m) { g_create_error = "shm selftest: MAX mismatch"; return SPHMI_ERR_STATE; }
        }
        // neighbour exchange: two rounds, n_bytes per direction (longer than a ring when the caller asks for it)
        auto pat = [](int from, int to, int round, size_t i) { return (char)((i * 31 + (size_t)from * 7 + (size_t)to * 3 + (size_t)round) & 0xff); };
        for (int round = 0; round < 2; ++round) {
            const size_t nb = (size_t)n_bytes + (size_t)round;
            std::vector<char> sl(nb), sr(nb), rl(nb), rr(nb);
            for (size_t i = 0; i < nb; ++i) { sl[i] = pat(rank, rank - 1, round, i); sr[i] = pat(rank, rank + 1, round, i); }
            std::vector<ShmWorld::Xfer> x;
            if (rank > 0) { x.push_back({true, 0, sl.data(), nb, 0}); x.push_back({false, 0, rl.data(), nb, 0}); }
            if (rank < world - 1) { x.push_back({true, 1, sr.data(), nb, 0}); x.push_back({false, 1, rr.data(), nb, 0}); }
            w.exchange(x);
            for (size_t i = 0; i < nb; ++i) {
                if (rank > 0 && rl[i] != pat(rank - 1, rank, round, i)) { g_create_error = "shm selftest: message from the left differs"; return SPHMI_ERR_STATE; }
                if (rank < world - 1 && rr[i] != pat(rank + 1, rank, round, i)) { g_create_error = "shm selftest: message from the right differs"; return SPHMI_ERR_STATE; }
            }
        }
        w.barrier();
        return SPHMI_OK;
    } catch (const std::exception& x) { g_create_error = x.what(); return SPHMI_ERR_DEVICE; }
}
int sphmi_dam_break_3d_count(double dp, int64_t* n_bound_out, int64_t* n_fluid_out) {
    if (!(dp > 0)) return SPHMI_ERR_ARGUMENT;
    auto rnd = [](double x) { return (long long)std::floor(x + 0.5); };
    const long long nx = rnd(1.6 / dp) + 1, ny = rnd(0.66 / dp) + 1, kwall = rnd(0.40 / dp), kcap = rnd(0.44 / dp);
    const long long px = rnd(0.12 / dp) + 1, py = rnd(0.14 / dp) + 1;
    auto perim = [](long long a, long long b) { return 2 * (a + b) - 4; };
    if (n_bound_out) *n_bound_out = nx * ny - (px - 2) * (py - 2) + kwall * perim(nx, ny) + (kcap - 1) * perim(px, py) + px * py;
    if (n_fluid_out) *n_fluid_out = (rnd(0.38 / dp) + 1) * (rnd(0.62 / dp) + 1) * (rnd(0.28 / dp) + 1);
    return SPHMI_OK;
}
int sphmi_generate_dam_break_3d(sphmi_handle* h, double dp) { SPHMI_GUARD(h, h->e->generate_dam_break_3d(dp)); }
int sphmi_owned_count(sphmi_handle* h, int64_t* n_out) {
    if (!n_out) return SPHMI_ERR_ARGUMENT;
    SPHMI_GUARD(h, *n_out = h->e->owned_count());
}
int sphmi_multi_info_get(sphmi_handle* h, sphmi_multi_info* out) {
    if (!out) return SPHMI_ERR_ARGUMENT;
    SPHMI_GUARD(h, ([&] {
        if (auto* m = dynamic_cast<sphmi::MultiEngine<float>*>(h->e)) m->multi_info(out);
        else if (auto* d = dynamic_cast<sphmi::MultiEngine<double>*>(h->e)) d->multi_info(out);
        else { memset(out, 0, sizeof *out); out->world = 1; out->n_local = 1; }
    }()));
}
int sphmi_multi_set_cuts(sphmi_handle* h, const int64_t* cuts, int32_t n) {
    if (!cuts || n < 0 || n >= SPHMI_MAX_DEVICES) return SPHMI_ERR_ARGUMENT;
    SPHMI_GUARD(h, ([&] {
        sphmi::SlabPlan p;
        p.lo.assign(n + 1, 0); p.hi.assign(n + 1, 0);
        for (int r = 0; r <= n; ++r) { p.lo[r] = r == 0 ? -sphmi::SlabPlan::INF : cuts[r - 1]; p.hi[r] = r == n ? sphmi::SlabPlan::INF : cuts[r] - 1; }
        if (auto* m = dynamic_cast<sphmi::MultiEngine<float>*>(h->e)) m->given_plan = p;
        else if (auto* d = dynamic_cast<sphmi::MultiEngine<double>*>(h->e)) d->given_plan = p;
        else throw sphmi::EngineError(SPHMI_ERR_STATE, "sphmi_multi_set_cuts: not a multi-device handle");
    }()));
}
int sphmi_multi_column_cost(sphmi_handle* h, int64_t col0, int32_t ncols, uint64_t* cost_out) {
    if (!cost_out || ncols < 1) return SPHMI_ERR_ARGUMENT;
    SPHMI_GUARD(h, ([&] {
        if (auto* m = dynamic_cast<sphmi::MultiEngine<float>*>(h->e)) m->column_cost(col0, ncols, cost_out);
        else if (auto* d = dynamic_cast<sphmi::MultiEngine<double>*>(h->e)) d->column_cost(col0, ncols, cost_out);
        else throw sphmi::EngineError(SPHMI_ERR_STATE, "sphmi_multi_column_cost: not a multi-device handle");
    }()));
}
int sphmi_multi_halo_info(sphmi_handle* h, int64_t* out, int32_t capacity_words, int32_t* n_words_out) {
    if (!out || !n_words_out || capacity_words < 12) return SPHMI_ERR_ARGUMENT;
    SPHMI_GUARD(h, ([&] {
        if (auto* m = dynamic_cast<sphmi::MultiEngine<float>*>(h->e)) *n_words_out = m->halo_info(out, capacity_words);
        else if (auto* d = dynamic_cast<sphmi::MultiEngine<double>*>(h->e)) *n_words_out = d->halo_info(out, capacity_words);
        else throw sphmi::EngineError(SPHMI_ERR_STATE, "sphmi_multi_halo_info: not a multi-device handle");
    }()));
}
int sphmi_plan_slabs(const sphmi_config* cfg, const void* position, const void* ghost_points, int64_t n, int32_t world,
                     int32_t* axis_out, int32_t* halo_width_out, int64_t* cuts_out, int64_t* owned_out, int64_t* capacity_out) {
    using namespace sphmi;
    if (!cfg || !position || n < 1 || world < 1 || world > SPHMI_MAX_DEVICES) return SPHMI_ERR_ARGUMENT;
    try {
        SlabSetup S;
        plan_slabs(*cfg, position, ghost_points, n, world, cfg->slab_axis - 1, nullptr, 1.6, S);
        if (axis_out) *axis_out = S.axis;
        if (halo_width_out) *halo_width_out = S.halo_width;
        if (cuts_out) for (int r = 1; r < world; ++r) cuts_out[r - 1] = S.plan.lo[r];
        if (owned_out) { for (int r = 0; r < world; ++r) owned_out[r] = 0; for (int o : S.owner) owned_out[o] += 1; }
        if (capacity_out) for (int r = 0; r < world; ++r) capacity_out[r] = S.capacity[r];
        return SPHMI_OK;
    } catch (const EngineError& x) { g_create_error = x.what(); return x.status; }
    catch (const std::exception& x) { g_create_error = x.what(); return SPHMI_ERR_ARGUMENT; }
}

static int check_config(const sphmi_config* cfg, int slabs, std::string& why) {
    auto fail = [&](int st, const std::string& m) { why = m; return st; };
    if (cfg->struct_size != (int32_t)sizeof(sphmi_config) || cfg->abi_version != SPHMI_ABI_VERSION)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: struct_size / abi_version mismatch");
    if (cfg->dims != 2 && cfg->dims != 3) return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: dims must be 2 or 3");
    if ((cfg->host_float_bytes != 4 && cfg->host_float_bytes != 8) ||
        (cfg->device_float_bytes != 4 && cfg->device_float_bytes != 8))
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: float_bytes must be 4 or 8");
    // the neighbour gathers use 32-bit buffer offsets: n × sizeof(packet) must stay below 4 GB
    // (n × sizeof(packet) < 4 GB: 2^27 packets of 16 bytes, 2^26 of 32; a multi-device handle holds n / devices per slab)
    {
        const int64_t per = slabs > 1 ? (cfg->n_particles + slabs - 1) / slabs : cfg->n_particles;
        const int64_t lim = (cfg->device_float_bytes == 8 ? (1ll << 26) : (1ll << 27)) - 1;     // N × record size (64 / 32 bytes) < 4 GB
        if (cfg->n_particles < 1 || per > lim)
            return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: n_particles out of range [1, 2^27) (fp32 kernels) / [1, 2^26) (fp64 kernels) per device");
    }
    if (cfg->kernel != SPHMI_KERNEL_WENDLAND_C2 && cfg->kernel != SPHMI_KERNEL_CUBIC_SPLINE)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: kernel not implemented");
    if (cfg->kernel_output != SPHMI_KOUT_NONE && cfg->kernel_output != SPHMI_KOUT_STORE)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: kernel output mode not implemented");
    if (cfg->viscosity < SPHMI_VISC_ZERO || cfg->viscosity > SPHMI_VISC_LAMINAR_SPS)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: viscosity model not implemented");
    if (cfg->density_diffusion < SPHMI_DDT_NONE || cfg->density_diffusion > SPHMI_DDT_COMPLEX)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: density diffusion model not implemented");
    if (cfg->shifting != SPHMI_SHIFT_NONE && cfg->shifting != SPHMI_SHIFT_PLANAR)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: shifting mode not implemented");
    if (cfg->mdbc != SPHMI_MDBC_NONE && cfg->mdbc != SPHMI_MDBC_SIMPLE)
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: mdbc mode not implemented");
    if (!(cfg->h > 0) || !(cfg->H > 0) || !(cfg->rho0 > 0) || !(cfg->m0 > 0) || !(cfg->c0 > 0) || !(cfg->CFL > 0))
        return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: h, H, rho0, m0, c0, CFL must be positive");
    if (cfg->n_devices < 0 || cfg->n_devices > SPHMI_MAX_DEVICES) return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: n_devices out of range [0, 16]");
    if (cfg->slab_axis < 0 || cfg->slab_axis > cfg->dims) return fail(SPHMI_ERR_ARGUMENT, "sphmi_create: slab_axis must be 0 (auto) or 1 … dims");
    return SPHMI_OK;
}

int sphmi_destroy(sphmi_handle* h) {
    if (!h) return SPHMI_OK;
    delete h->e;
    delete h;
    return SPHMI_OK;
}

int sphmi_upload(sphmi_handle* h, const void* position, const void* velocity, const void* acceleration,
                 const void* density, const uint8_t* type, const int64_t* id, const uint64_t* group_marker,
                 const void* ghost_points) {
    SPHMI_GUARD(h, (h->e->reset_count(), h->e->upload(position, velocity, acceleration, density, type, id, group_marker, ghost_points)));
}

int sphmi_download_begin(sphmi_handle* h, void* position, void* velocity, void* acceleration, void* density,
                         void* pressure, int64_t* id, uint8_t* type, uint64_t* group_marker, void* ghost_points,
                         int64_t* cells) {
    SPHMI_GUARD(h, h->e->download_begin(position, velocity, acceleration, density, pressure, id, type, group_marker,
                                        ghost_points, cells));
}
int sphmi_set_output_components(sphmi_handle* h, int components) { SPHMI_GUARD(h, h->e->set_output_components(components)); }
int sphmi_download_end(sphmi_handle* h) { SPHMI_GUARD(h, h->e->download_end()); }
int sphmi_host_register(sphmi_handle* h, void* ptr, int64_t bytes) { SPHMI_GUARD(h, h->e->host_register(ptr, (size_t)bytes)); }
int sphmi_host_unregister(sphmi_handle* h, void* ptr) { SPHMI_GUARD(h, h->e->host_unregister(ptr)); }
int sphmi_download_kernel_output(sphmi_handle* h, void* kernel, void* kernel_gradient) {
    SPHMI_GUARD(h, h->e->download_kernel_output(kernel, kernel_gradient));
}
int sphmi_download_permutation(sphmi_handle* h, int64_t* prev_row) { SPHMI_GUARD(h, h->e->download_permutation(prev_row)); }
int sphmi_set_motion(sphmi_handle* h, uint64_t group_marker, double velocity, double start_time, double duration,
                     const double* direction) {
    SPHMI_GUARD(h, h->e->set_motion(group_marker, velocity, start_time, duration, direction));
}
int sphmi_set_clock(sphmi_handle* h, int64_t iteration, double total_time) {
    SPHMI_GUARD(h, (h->e->iteration = iteration, h->e->total_time = total_time));
}

int sphmi_advance(sphmi_handle* h, double t_target, int64_t max_steps, sphmi_progress* out) {
    SPHMI_GUARD(h, h->e->advance(t_target, max_steps, out));
}

int sphmi_download(sphmi_handle* h, void* position, void* velocity, void* acceleration, void* density,
                   void* pressure, int64_t* id, uint8_t* type, uint64_t* group_marker, void* ghost_points,
                   int64_t* cells) {
    SPHMI_GUARD(h, h->e->download(position, velocity, acceleration, density, pressure, id, type, group_marker,
                                  ghost_points, cells));
}

int sphmi_forces_once(sphmi_handle* h, int apply_mdbc, void* drhodt, void* acceleration) {
    SPHMI_GUARD(h, h->e->forces_once(apply_mdbc, drhodt, acceleration));
}

int sphmi_unique_cells(sphmi_handle* h, int64_t* cells_out, int64_t capacity, int64_t* n_out) {
    SPHMI_GUARD(h, h->e->unique_cells(cells_out, capacity, n_out));
}

int sphmi_timers(sphmi_handle* h, int32_t capacity, const char** names_out, double* seconds_out,
                 int64_t* calls_out, int32_t* n_out) {
    SPHMI_GUARD(h, h->e->timers(capacity, names_out, seconds_out, calls_out, n_out));
}

int sphmi_force_kernel_stats(sphmi_handle* h, int reset, double* avg_ms_out, int64_t* launches_out) {
    SPHMI_GUARD(h, h->e->force_stats(reset, avg_ms_out, launches_out));
}

int sphmi_device_ptrs(sphmi_handle* h, void** pk0, void** pk1, int64_t* n_local) {
    SPHMI_GUARD(h, h->e->device_ptrs(pk0, pk1, n_local));
}

}  // extern "C"
