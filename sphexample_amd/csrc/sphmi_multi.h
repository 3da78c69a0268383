// sphmi_multi.h — the slab driver INSIDE libsphmi.so: one sphmi_create / sphmi_advance drives the GPUs of a node.
//
// The reference has no multi-device path; its caller runs ONE host process and calls
// SimulationLoop(...) once per output interval (src/SPHCellList.jl:727-805, called at :883).  This file keeps that
// contract: a handle created with a device list (sphmi_config.n_devices > 1) owns one slab engine per device and
// sphmi_advance runs the whole interval — reductions → MAX-allreduce → device-side step control → halo ‖ interior
// tiles → edge tiles, twice per step — from ONE host thread, with no host round trip inside a batch of queued steps.
// A second way in, sphmi_create_rank, is the same driver with ONE local slab per process and the peers behind RCCL
// (what `torchrun bench.py --gpus N` uses: the contract there is one process per GPU).
//
// Decomposition (unchanged from round 1, now host C++): 1-D slabs cut on cell-column boundaries by WORK (candidates
// in the 3^D cells around a particle, the measure of the tile schedule), exact lightest-heaviest-slab cuts (dynamic
// programme), slab axis chosen for balance, ghost layer = one cell column per side (2 + off with mDBC) kept as
// ordinary entries of the sorted arrays, migration + re-cut at the collective rebuild, 64-bit order tags that keep
// the reference's in-cell order independent of the slab a particle lives in.
//
// Transport: RCCL (ncclSend / ncclRecv between slab neighbours — each a direct xGMI link — and one ncclAllReduce of
// four uint64 bit patterns per step), bound at run time from librccl.so.1; or, when the ranks of ONE process share a
// device (the single-GPU test configuration) or SPHMI_TRANSPORT=local, stream-ordered device copies; or, for
// sphmi_create_rank with SPHMI_TRANSPORT=shm, host staging through a shared-memory segment (sphmi_shm.h: processes that
// share a GPU — RCCL refuses that — run the rank-mode driver unchanged; a bring-up / test transport, nothing overlaps).
#pragma once
#include <dlfcn.h>

#include <array>
#include <chrono>
#include <functional>
#include <memory>
#include <numeric>
#include <unordered_map>

#include "sphmi_shm.h"

namespace sphmi {

// ------------------------------------------------------------------------------------------------------------------
// device helpers of the collective rebuild: predicates → ascending index lists (stable compaction)
// ------------------------------------------------------------------------------------------------------------------
// flag[i] = ((type[i] & tmask) == tval) && lo <= cx[i] <= hi      (type 0 = dead never matches: tval != 0 or tmask == 0 handled by `live`)
__global__ void __launch_bounds__(256) k_dd_flag(const int* cx, const uint8_t* type, int N, int tmask, int tval, int any_live,
                                                 long long lo, long long hi, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int t = type[i];
    const bool tm = any_live ? (t != 0) : ((t & tmask) == tval && t != 0);
    const long long c = cx[i];
    flag[i] = (tm && c >= lo && c <= hi) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_dd_compact(const int* flag, const int* pos, int N, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && flag[i]) out[pos[i]] = i;
}
// min / max of cx over the listed particles (slab-skip check) and over owned particles (re-cut extent)
// (grid-stride over at most 512 blocks: one pair of atomics per WAVE of a 1 M-row launch — 16 k of them on one cache line — was 0.19–0.37 ms)
__global__ void __launch_bounds__(256) k_dd_minmax(const int* cx, const uint8_t* type, const int* idx, int n, int* mm) {
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int i = idx ? idx[k] : k;
        if (idx || (type[i] != 0 && !(type[i] & kGhostMask))) { const int c = cx[i]; lo = min(lo, c); hi = max(hi, c); }
    }
    __shared__ int s_lo[4], s_hi[4];
    lo = wave_min_i(lo); hi = wave_max_i(hi);
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = min(lo, s_lo[w]); hi = max(hi, s_hi[w]); }
        if (lo != INT32_MAX) atomicMin(&mm[0], lo);
        if (hi != INT32_MIN) atomicMax(&mm[1], hi);
    }
}
// cells that hold owned particles (a cell is wholly owned or wholly ghost: the cuts run along cell columns)
__global__ void __launch_bounds__(256) k_dd_count_owned_cells(const int* key, const uint8_t* type, int N, int ncell, int* out) {
    // (grid-stride, one atomic per block: see k_dd_minmax)
    int n = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int k = key[i];
        n += (k < ncell && type[i] != 0 && !(type[i] & kGhostMask) && (i == 0 || key[i - 1] != k)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
    __shared__ int s_n[4];
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) { const int t = s_n[0] + s_n[1] + s_n[2] + s_n[3]; if (t) atomicAdd(out, t); }
}
// a stream-ordered copy between two buffers this process can address — the slabs of one handle on one device, or on devices with
// peer access: `hipMemcpyAsync` device → device moved a 4.4 MB halo at ≈40 GB/s (110 µs a message, four messages per step and slab:
// a third of the step of two slabs sharing a GPU — profiles/r04_slab_overhead.md)
__global__ void __launch_bounds__(256) k_dd_copy(const void* src, void* dst, size_t bytes) {
    const size_t n16 = bytes >> 4;
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(dst);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) d[k] = s[k];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) reinterpret_cast<char*>(dst)[(n16 << 4) + threadIdx.x] = reinterpret_cast<const char*>(src)[(n16 << 4) + threadIdx.x];
}
struct RedPtrs { const unsigned long long* p[16]; int n; };
// The per-step decisions of a slab (one wave): M = max(M, T_0, …) over four uint64 bit patterns — T_0 this slab's slots after the
// allreduce (or, between the slabs of one process, every slab's own slots) — then the SAME step_control_decide as the
// one-device engine on M; M is zeroed when the step consumed it, and `zero_set`, the OTHER of the slab engine's two sets of
// reduction slots, is zeroed for the corrector of the coming step.  One launch where round 2 had three (take, merge, control):
// the chain corrector → [allreduce] → control → predictor is the critical path of every step on every GPU.
template <class T>
__global__ void k_dd_merge_control(unsigned long long* M, RedPtrs Tp, unsigned long long* zero_set, StepCtrl* cp, double h, double c0, double CFL) {
    const int i = threadIdx.x;
    unsigned long long v = 0;
    if (i < 4) {
        v = M[i];
        for (int k = 0; k < Tp.n; ++k) { const unsigned long long u = Tp.p[k][i]; v = u > v ? u : v; }
    }
    const unsigned long long r0 = __shfl(v, 0, 64), r1 = __shfl(v, 1, 64), r2 = __shfl(v, 2, 64), r3 = __shfl(v, 3, 64);
    int consumed = 0;
    if (i == 0) {
        StepCtrl c = *cp;
        consumed = step_control_decide<T>(r0, r1, r2, r3, c, h, c0, CFL) ? 1 : 0;
        *cp = c;
    }
    consumed = __shfl(consumed, 0, 64);
    if (i < 4) { M[i] = consumed ? 0ull : v; zero_set[i] = 0ull; }
}
// The same WITHOUT a collective (round 5, $SPHMI_EXCHANGE=mailbox): every slab owns a MAILBOX in its device memory — [2 parities][16 senders][8 words]:
// { sequence number, four maxima } — that its peers can address (peer access between the devices of one process, hipIpc between processes).  The control
// launch of a step first POSTS this slab's four maxima into every slab's box (lane q serves slab q: four relaxed stores, a release fence, the sequence
// number — all at system scope), then lane q WAITS for slab q's post of this step in its OWN box and the wave takes the maximum: one launch, a handful of
// stores over xGMI and one load round trip on the critical path corrector → control → predictor, where ncclAllReduce of 32 bytes pays a collective's
// latency.  Two parities: a peer cannot post step n + 2 before this slab has read step n (it needs this slab's post of step n + 1 first).  The same four
// words travel and the reduction is an integer maximum, so the result is the allreduce's bit for bit.  A peer that never posts is an error after
// `timeout` ticks of the 100-MHz clock (StepCtrl::error = 4), not a hang.
struct MboxArgs {
    unsigned long long* dst[16];   // every slab's box as THIS device addresses it (null: no such slab)
    const unsigned long long* own; // this slab's box
    const unsigned long long* mine;// the four slots the last corrector of this slab filled
    int world, me, parity;
    unsigned long long seq, timeout;
};
constexpr int kMboxWords = 2 * 16 * 8;
template <class T>
__global__ void k_dd_mbox_merge_control(unsigned long long* M, MboxArgs A, unsigned long long* zero_set, StepCtrl* cp, double h, double c0, double CFL) {
    const int i = threadIdx.x;
    const size_t slot = ((size_t)A.parity * 16) * 8;
    if (i < A.world && A.dst[i] != nullptr) {
        unsigned long long* b = A.dst[i] + slot + (size_t)A.me * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) __hip_atomic_store(b + 1 + k, A.mine[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(b, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned long long v[4] = {0ull, 0ull, 0ull, 0ull};
    int late = 0;
    if (i < A.world) {
        const unsigned long long* b = A.own + slot + (size_t)i * 8;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != A.seq) {
            __builtin_amdgcn_s_sleep(4);
            if ((unsigned long long)__builtin_amdgcn_s_memrealtime() - t0 > A.timeout) { late = 1; break; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load(b + 1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)v[k], o, 64), hi = __shfl_xor((unsigned)(v[k] >> 32), o, 64);
            const unsigned long long u = ((unsigned long long)hi << 32) | lo;
            v[k] = u > v[k] ? u : v[k];
        }
    late = __builtin_amdgcn_ballot_w64(late != 0) != 0;
    if (i == 0) {
        unsigned long long r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned long long m = M[k]; r[k] = m > v[k] ? m : v[k]; }
        StepCtrl c = *cp;
        bool consumed = false;
        if (late) { c.error = 4; c.active = 0; }
        else consumed = step_control_decide<T>(r[0], r[1], r[2], r[3], c, h, c0, CFL);
        *cp = c;
#pragma unroll
        for (int k = 0; k < 4; ++k) { M[k] = consumed ? 0ull : r[k]; zero_set[k] = 0ull; }
    }
}
// both halo lists of a side in one launch: [0, nl) → buf_l, [nl, nl + nr) → buf_r
template <class T>
__global__ void __launch_bounds__(256) k_halo_pack2(Half<const typename Vec4<T>::type> pk0, Half<const typename Vec4<T>::type> pk1,
                                                    const int* idx_l, int nl, typename Vec4<T>::type* buf_l,
                                                    const int* idx_r, int nr, typename Vec4<T>::type* buf_r) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nl) { const int i = idx_l[k]; buf_l[k] = pk0[i]; buf_l[nl + k] = pk1[i]; return; }
    k -= nl;
    if (k < nr) { const int i = idx_r[k]; buf_r[k] = pk0[i]; buf_r[nr + k] = pk1[i]; }
}
template <class T>
__global__ void __launch_bounds__(256) k_halo_unpack2(Half<typename Vec4<T>::type> pk0, Half<typename Vec4<T>::type> pk1,
                                                      const int* idx_l, int nl, const typename Vec4<T>::type* buf_l,
                                                      const int* idx_r, int nr, const typename Vec4<T>::type* buf_r) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nl) { const int i = idx_l[k]; pk0[i] = buf_l[k]; pk1[i] = buf_l[nl + k]; return; }
    k -= nl;
    if (k < nr) { const int i = idx_r[k]; pk0[i] = buf_r[k]; pk1[i] = buf_r[nr + k]; }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side planning (pure C++: also reachable without a device through sphmi_plan_slabs, for the CPU tests)
// ------------------------------------------------------------------------------------------------------------------
struct SlabPlan {
    static constexpr long long INF = 1ll << 30;
    std::vector<long long> lo, hi;     // rank r owns the cell columns lo[r] … hi[r] (inclusive)
    int min_width = 2;                 // columns a slab keeps: at least the halo width (a ghost layer comes from ONE neighbour)
    int world() const { return (int)lo.size(); }
    std::vector<long long> cuts() const { std::vector<long long> c; for (int r = 1; r < world(); ++r) c.push_back(lo[r]); return c; }
    int owner_of(long long cx) const { int r = 0; while (r + 1 < world() && cx >= lo[r + 1]) ++r; return r; }
};

// map_floor (src/SPHCellList.jl:56-61) in the arithmetic of the DEVICE type: the ownership the host assigns must be
// the cell the kernels will compute
inline long long host_cell(double x, double H_inv, int device_float_bytes) {
    if (device_float_bytes == 4) { const float xf = (float)x; const float t = truncf(fmaf(fabsf(xf), (float)H_inv, 0.5f)); return xf < 0 ? -(long long)t : (long long)t; }
    const double t = std::trunc(std::fma(std::fabs(x), H_inv, 0.5));
    return x < 0 ? -(long long)t : (long long)t;
}

// cut positions c_1 < … < c_{world-1} over columns 0 … n-1 that MINIMISE THE HEAVIEST SLAB, every slab at least `width`
// columns, cut r within [lo_b[r], hi_b[r]] when given.  Exact (dynamic programme).  false: infeasible.
inline bool best_cuts(const std::vector<double>& hist, int world, int width, const std::vector<long long>* lo_b,
                      const std::vector<long long>* hi_b, std::vector<long long>& cuts) {
    const int n = (int)hist.size();
    std::vector<double> cum(n + 1, 0.0);
    for (int i = 0; i < n; ++i) cum[i + 1] = cum[i] + hist[i];
    const double BIG = std::numeric_limits<double>::infinity();
    std::vector<double> f(n + 1, BIG), g;
    f[0] = 0.0;
    std::vector<std::vector<int>> back(world, std::vector<int>(n + 1, 0));
    for (int r = 1; r <= world; ++r) {
        g.assign(n + 1, BIG);
        long long c0 = r == world ? n : std::max<long long>((long long)r * width, lo_b ? (*lo_b)[r] : 0);
        long long c1 = r == world ? n : std::min<long long>(n - (long long)(world - r) * width, hi_b ? (*hi_b)[r] : n);
        for (long long c = c0; c <= c1; ++c) {
            double best = BIG; int arg = 0;
            for (long long p = 0; p <= c - width; ++p) {
                const double v = std::max(f[p], cum[c] - cum[p]);
                if (v < best) { best = v; arg = (int)p; }
            }
            g[c] = best; back[r - 1][c] = arg;
        }
        f = g;
    }
    if (!(f[n] < BIG)) return false;
    cuts.assign(world - 1, 0);
    int c = n;
    for (int r = world; r >= 1; --r) { c = back[r - 1][c]; if (r >= 2) cuts[r - 2] = c; }
    return true;
}

inline bool plan_from_hist(long long col0, const std::vector<double>& hist, int world, int min_width, SlabPlan& plan) {
    std::vector<long long> inner;
    if (!best_cuts(hist, world, world > 1 ? min_width : 1, nullptr, nullptr, inner)) return false;
    plan.lo.assign(world, 0); plan.hi.assign(world, 0); plan.min_width = min_width;
    for (int r = 0; r < world; ++r) {
        plan.lo[r] = r == 0 ? -SlabPlan::INF : col0 + inner[r - 1];
        plan.hi[r] = r == world - 1 ? SlabPlan::INF : col0 + inner[r] - 1;
    }
    return true;
}
// best cuts for the CURRENT global column histogram, every cut between its two old neighbours (a particle changes rank
// by at most one: migration stays a neighbour exchange), every slab at least min_width columns
inline SlabPlan plan_recut(const SlabPlan& old, long long col0, const std::vector<double>& hist) {
    const int world = old.world(), w = old.min_width;
    const long long n = (long long)hist.size();
    std::vector<long long> oc(world + 1);
    oc[0] = col0; oc[world] = col0 + n;
    for (int r = 1; r < world; ++r) oc[r] = old.lo[r];
    std::vector<long long> lo_b(world + 1, 0), hi_b(world + 1, 0), inner;
    for (int r = 1; r < world; ++r) { lo_b[r] = oc[r - 1] + w - col0; hi_b[r] = oc[r + 1] - w - col0; }
    lo_b[world] = hi_b[world] = n;
    SlabPlan p = old;
    if (!best_cuts(hist, world, w, &lo_b, &hi_b, inner)) return p;
    for (int r = 0; r < world; ++r) {
        p.lo[r] = r == 0 ? -SlabPlan::INF : col0 + inner[r - 1];
        p.hi[r] = r == world - 1 ? SlabPlan::INF : col0 + inner[r] - 1;
    }
    return p;
}

struct SlabSetup {
    int axis = 0, halo_width = 1;
    SlabPlan plan;
    std::vector<int> owner;              // rank of every particle
    std::vector<long long> capacity;     // particles a rank's handle must hold
};

// cols[a][i]: cell column of particle i along axis a.  work[i] = candidates in the 3^D cells around particle i.
inline std::vector<double> particle_work(const std::vector<std::vector<long long>>& cols) {
    const int D = (int)cols.size();
    const size_t N = cols[0].size();
    long long lo[3] = {0, 0, 0}, dim[3] = {1, 1, 1};
    for (int a = 0; a < D; ++a) {
        long long mn = cols[a][0], mx = cols[a][0];
        for (size_t i = 1; i < N; ++i) { mn = std::min(mn, cols[a][i]); mx = std::max(mx, cols[a][i]); }
        lo[a] = mn; dim[a] = mx - mn + 3;                                  // one cell of padding per side
    }
    const size_t nc = (size_t)(dim[0] * dim[1] * dim[2]);
    std::vector<long long> grid(nc, 0), box(nc, 0);
    std::vector<size_t> lin(N);
    for (size_t i = 0; i < N; ++i) {
        size_t l = 0;
        for (int a = D - 1; a >= 0; --a) l = l * (size_t)dim[a] + (size_t)(cols[a][i] - lo[a] + 1);
        lin[i] = l; grid[l] += 1;
    }
    long long stride = 1;
    for (int a = 0; a < D; ++a) {                                          // separable 3-wide box sum (padding cells are empty)
        for (size_t l = 0; l < nc; ++l) {
            long long v = grid[l];
            const long long ca = (long long)(l / (size_t)stride) % dim[a];
            if (ca > 0) v += grid[l - (size_t)stride];
            if (ca + 1 < dim[a]) v += grid[l + (size_t)stride];
            box[l] = v;
        }
        grid.swap(box);
        stride *= dim[a];
    }
    std::vector<double> w(N);
    for (size_t i = 0; i < N; ++i) w[i] = (double)grid[lin[i]];
    return w;
}

// Slab axis, cuts, ownership and per-rank capacity for `world` ranks.  axis_req < 0: choose the axis whose best cuts
// leave the lightest heaviest rank; axes within 1 % tie → the thinnest ghost layers, then the slowest sort axis.
inline void plan_slabs(const sphmi_config& cfg, const void* position, const void* ghost_points, int64_t N, int world,
                       int axis_req, const SlabPlan* given, double capacity_factor, SlabSetup& out) {
    const int D = cfg.dims;
    auto coord = [&](const void* base, int64_t i, int a) -> double {
        return cfg.host_float_bytes == 8 ? ((const double*)base)[i * D + a] : (double)((const float*)base)[i * D + a];
    };
    std::vector<std::vector<long long>> cols(D, std::vector<long long>((size_t)N));
    for (int a = 0; a < D; ++a)
        for (int64_t i = 0; i < N; ++i) cols[a][(size_t)i] = host_cell(coord(position, i, a), cfg.H_inv, cfg.device_float_bytes);
    // halo width per axis: one column for the pair forces; with mDBC 2 + off, off = the widest column distance between a
    // boundary particle and its ghost node under any rounding within ±1e-4 of a cell (lattices sit exactly on cell edges)
    int widths[3] = {1, 1, 1};
    if (cfg.mdbc == SPHMI_MDBC_SIMPLE && ghost_points) {
        auto col = [](double u) { const double t = std::trunc(std::fabs(u) + 0.5); return (long long)(u < 0 ? -t : t); };
        auto dev = [&](double x) { return cfg.device_float_bytes == 4 ? (double)(float)x : x; };
        for (int a = 0; a < D; ++a) {
            long long off = 0;
            for (int64_t i = 0; i < N; ++i) {
                bool nz = false;
                for (int d = 0; d < D; ++d) nz |= coord(ghost_points, i, d) != 0.0;
                if (!nz) continue;
                const double ug = dev(coord(ghost_points, i, a)) * cfg.H_inv, ux = dev(coord(position, i, a)) * cfg.H_inv;
                const double e = 1e-4;
                off = std::max(off, std::max(std::llabs(col(ug + e) - col(ux - e)), std::llabs(col(ug - e) - col(ux + e))));
            }
            widths[a] = 2 + (int)off;
        }
    }
    std::vector<double> work;
    if (world > 1) work = particle_work(cols);
    auto hist_of = [&](int a, long long& c0) {
        long long mn = cols[a][0], mx = cols[a][0];
        for (auto c : cols[a]) { mn = std::min(mn, c); mx = std::max(mx, c); }
        std::vector<double> h((size_t)(mx - mn + 1), 0.0);
        for (size_t i = 0; i < (size_t)N; ++i) h[(size_t)(cols[a][i] - mn)] += work.empty() ? 1.0 : work[i];
        c0 = mn;
        return h;
    };
    int axis = axis_req;
    if (axis < 0) {
        struct Cand { double load; long long edge; int ax; };
        std::vector<Cand> cand;
        for (int a = 0; a < D; ++a) {
            long long c0; auto h = hist_of(a, c0);
            SlabPlan p;
            if (!plan_from_hist(c0, h, world, std::max(2, widths[a]), p)) continue;
            std::vector<double> load(world, 0.0);
            long long edge = 0;
            for (size_t i = 0; i < (size_t)N; ++i) {
                const int r = p.owner_of(cols[a][i]);
                load[r] += work.empty() ? 1.0 : work[i];
                for (int q = 1; q < world; ++q) edge += (cols[a][i] == p.lo[q] - 1 || cols[a][i] == p.lo[q]) ? 1 : 0;
            }
            cand.push_back({*std::max_element(load.begin(), load.end()), edge, a});
        }
        if (cand.empty()) throw EngineError(SPHMI_ERR_ARGUMENT, "no axis has enough cell columns per device: too many devices for this domain");
        double best = cand[0].load;
        for (auto& c : cand) best = std::min(best, c.load);
        const Cand* pick = nullptr;
        for (auto& c : cand) {
            if (c.load > 1.01 * best) continue;
            if (!pick || c.edge < pick->edge || (c.edge == pick->edge && c.ax > pick->ax)) pick = &c;
        }
        axis = pick->ax;
    }
    if (axis >= D) throw EngineError(SPHMI_ERR_ARGUMENT, "slab axis out of range");
    out.axis = axis; out.halo_width = widths[axis];
    long long c0; auto h = hist_of(axis, c0);
    if (given) out.plan = *given;
    else if (!plan_from_hist(c0, h, world, std::max(2, widths[axis]), out.plan))
        throw EngineError(SPHMI_ERR_ARGUMENT, "a slab would be narrower than the halo: too many devices for this domain");
    out.plan.min_width = std::max(out.plan.min_width, widths[axis]);
    out.owner.resize((size_t)N);
    std::vector<long long> n_own(world, 0);
    for (size_t i = 0; i < (size_t)N; ++i) { out.owner[i] = out.plan.owner_of(cols[axis][i]); n_own[out.owner[i]] += 1; }
    // capacity: owned + the ghost columns, with room for the fluid to pile up and for the cuts to move
    std::vector<long long> cnt(h.size(), 0);
    for (size_t i = 0; i < (size_t)N; ++i) cnt[(size_t)(cols[axis][i] - c0)] += 1;
    const long long lo_c = c0, hi_c = c0 + (long long)h.size() - 1;
    auto colc = [&](long long c) -> long long { return (c >= lo_c && c <= hi_c) ? cnt[(size_t)(c - lo_c)] : 0; };
    const long long hmax = *std::max_element(cnt.begin(), cnt.end());
    out.capacity.assign(world, 0);
    const int W = out.halo_width;
    for (int r = 0; r < world; ++r) {
        const long long s_lo = std::max(out.plan.lo[r], lo_c), s_hi = std::min(out.plan.hi[r], hi_c);
        long long ghosts = 0;
        for (int k = 1; k <= W; ++k) ghosts += colc(s_lo - k) + colc(s_hi + k);
        const long long slack = colc(s_lo - W - 1) + colc(s_hi + W + 1) + 2 * hmax;
        long long cap = (long long)(capacity_factor * (double)(n_own[r] + ghosts)) + slack + 1024;
        cap = std::max(cap, (long long)(capacity_factor * ((double)N / world)));
        out.capacity[r] = cap;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time (librccl.so.1: the copy already in the process — torch ships one — or /opt/rocm/lib's)
// ------------------------------------------------------------------------------------------------------------------
// The handful of NCCL / RCCL declarations the driver binds (the NCCL 2.x C ABI; values as in <rccl/rccl.h>).  Declared here
// so that libsphmi.so builds on a ROCm install without the RCCL development headers: RCCL is needed at RUN time only, and
// only by handles that spread over several devices.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
}
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;      // optional (newer RCCL): the text behind a bare "invalid usage"
    // Bound once per process; a failed attempt leaves nothing behind (the table is published only when every symbol resolved),
    // so the next call tries again and fails with the same clean error instead of handing out null function pointers.
    static Rccl& get() {
        static Rccl ready;
        if (ready.so) return ready;
        Rccl r;
        std::string tried;
        const char* names[] = {getenv("SPHMI_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // SPHMI_RCCL_LIB is an override, not one more candidate: a wrong path must fail, not fall through to another copy
        const int n_names = (names[0] && *names[0]) ? 1 : 4;
        for (int k = 0; k < n_names && !r.so; ++k) {
            const char* n = names[k];
            if (!n || !*n) continue;
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!r.so) { const char* e = dlerror(); tried += std::string(tried.empty() ? "" : "; ") + (e ? e : n); }      // dlerror() clears itself: read ONCE
        }
        if (!r.so) throw EngineError(SPHMI_ERR_DEVICE, "RCCL not found (needed by multi-device handles): " + tried);
        auto sym = [&](const char* s, bool required = true) {
            void* p = dlsym(r.so, s);
            if (!p && required) throw EngineError(SPHMI_ERR_DEVICE, std::string("RCCL symbol missing: ") + s);
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetLastError = (decltype(r.GetLastError))sym("ncclGetLastError", false);
        ready = r;
        return ready;
    }
};
// What the failing call was doing: which slab, which peer, which phase — RCCL's own text is often a bare "invalid usage".
struct RcclCtx { int rank = -1, peer = -1; const char* what = ""; ncclComm_t comm = nullptr; };
static thread_local RcclCtx g_rccl_ctx;
inline std::string rccl_fail(const char* expr, ncclResult_t r_) {
    Rccl& N = Rccl::get();
    std::string m = std::string(expr) + ": " + N.GetErrorString(r_);
    if (N.GetLastError && g_rccl_ctx.comm) { const char* d = N.GetLastError(g_rccl_ctx.comm); if (d && *d) m += std::string(" — ") + d; }
    char buf[160];
    snprintf(buf, sizeof buf, " [%s; slab %d%s", g_rccl_ctx.what, g_rccl_ctx.rank, g_rccl_ctx.peer >= 0 ? "" : "]");
    m += buf;
    if (g_rccl_ctx.peer >= 0) { snprintf(buf, sizeof buf, ", peer slab %d]", g_rccl_ctx.peer); m += buf; }
    return m;
}
#define NC(expr)                                                                                                      \
    do {                                                                                                              \
        ncclResult_t r_ = (expr);                                                                                     \
        if (r_ != ncclSuccess) throw EngineError(SPHMI_ERR_DEVICE, rccl_fail(#expr, r_));                              \
    } while (0)
#define NCX(what_, rank_, peer_, comm_, expr) do { g_rccl_ctx = RcclCtx{rank_, peer_, what_, comm_}; NC(expr); } while (0)

// a device buffer that only grows (rebuild-time lists and message buffers)
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    void* need(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            const size_t want = bytes + bytes / 4 + 256;
            if (hipMalloc(&p, want) != hipSuccess) throw EngineError(SPHMI_ERR_DEVICE, "hipMalloc failed (slab driver buffer)");
            cap = want;
        }
        return p;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// a page-locked host buffer that only grows (staging of the shm transport)
struct HostBuf {
    char* p = nullptr; size_t cap = 0;
    char* need(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void**)&p, want) != hipSuccess) throw EngineError(SPHMI_ERR_DEVICE, "hipHostMalloc failed (shm transport staging)");
            cap = want;
        }
        return p;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// one point-to-point message of a phase (global ranks; buffers are device memory of the local end(s))
struct Msg { int src, dst; const void* sbuf; void* rbuf; size_t bytes; int slot; };   // slot: which persistent buffer pair (WAR tracking of the local transport)

template <class T> struct MultiEngine;

// ------------------------------------------------------------------------------------------------------------------
template <class T>
struct MultiEngine final : EngineBase {
    using V4 = typename Vec4<T>::type;
    struct Halo {                          // one per state set: [0] = A (halo_width columns), [1] = H (one column)
        DevBuf send_l, send_r, slot_l, slot_r, sb_l, sb_r, rb_l, rb_r;
        int n_send_l = 0, n_send_r = 0, n_slot_l = 0, n_slot_r = 0;
    };
    struct Rank {
        int rank = 0, device = 0;
        std::unique_ptr<Engine<T>> e;
        hipStream_t main = nullptr, side = nullptr;
        hipEvent_t ev_pack = nullptr, ev_edge = nullptr, ev_red = nullptr, ev_consumed[8] = {};
        bool consumed_valid[8] = {};
        Halo halo[2];
        // The travelling slots are the slab engine's own two sets of reduction slots (Engine::red_d, alternating by step parity: the
        // corrector of step n fills one, the control of step n+1 reads it — after the allreduce in place — and zeroes the other).
        unsigned long long *M = nullptr, *stage = nullptr;   // M: merged maxima not yet consumed by a step, stage: peers' slots (other devices)
        unsigned long long* box = nullptr;                   // $SPHMI_EXCHANGE=mailbox: this slab's mailbox, and every slab's as this device addresses it
        unsigned long long* box_of[16] = {};
        void* box_ipc[16] = {};                              // … those opened from another process's handle (closed with the engine)
        DevBuf cx, flag, pos, idx[4], rec_s[2], rec_r[2], cost;
        int* mm_d = nullptr; int* mm_h = nullptr;
        int64_t* cnt_h = nullptr;
        // Two communicators per slab: `comm` carries the point-to-point traffic (halos on the side stream, migration records
        // and counts on the main stream), `comm_red` the per-step allreduce on the main stream.  One communicator used from two
        // streams is legal only while every rank issues its calls in the same order and RCCL serialises them; with two, the
        // 32-byte allreduce of step n+1 never queues behind the halo of step n inside the library.  ($SPHMI_RCCL_ONE_COMM=1:
        // comm_red = comm, the round-2 arrangement.)
        ncclComm_t comm = nullptr, comm_red = nullptr;
        bool has_left = false, has_right = false;
    };
    int world = 1;                     // slabs in total
    std::vector<Rank> R;               // the LOCAL ones (all of them in one-process mode, one in rank mode)
    bool rank_mode = false;            // one local slab, peers in other processes
    bool use_rccl = false;
    bool peer_ok = true;               // every pair of this process's devices has peer access: the local transport copies with a kernel
    std::unique_ptr<ShmWorld> shm;     // rank mode with SPHMI_TRANSPORT=shm: peers behind a shared-memory segment instead of RCCL
    HostBuf stage_s, stage_r;
    unsigned long long* red_h = nullptr;
    int D = 0, axis = 0, halo_width = 1;
    SlabPlan plan;
    bool overlap = true, moving = false, have_halo = false;
    double recut_imbalance = 1.05; int64_t n_recuts = 0;
    int64_t n_total = 0;
    double dx_rate = 0.0;
    int parity = 0;
    bool mbox_on = false; unsigned long long mbox_seq = 0, mbox_timeout = 0;
    bool two_sorts = false;
    static constexpr int kBatch = 16;
    std::vector<int> dev_of;           // device of every global rank (one-process mode)

    Rank* local(int g) { for (auto& r : R) if (r.rank == g) return &r; return nullptr; }

    MultiEngine(const sphmi_config& c, int world_, int my_rank /* −1: all ranks local */, const void* unique_id) {
        cfg = c; D = c.dims; world = world_; out_comp = c.dims;
        rank_mode = my_rank >= 0;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw EngineError(SPHMI_ERR_DEVICE, "no HIP device available (libsphmi has no CPU fallback)");
        if (world < 1 || world > 16) throw EngineError(SPHMI_ERR_ARGUMENT, "1 to 16 slabs per handle");
        if (const char* w = getenv("SPHMI_DD_OVERLAP")) overlap = atoi(w) != 0;
        if (const char* w = getenv("SPHMI_DD_RECUT")) recut_imbalance = atof(w);
        if (const char* w = getenv("SPHMI_DD_TWO_SORTS")) two_sorts = atoi(w) != 0;
        if (const char* w = getenv("SPHMI_DD_ONE_SLAB_AT_A_TIME")) one_at_a_time = atoi(w) != 0;
        bool shared_device = false;
        if (rank_mode) {
            R.resize(1); R[0].rank = my_rank; R[0].device = c.device;
        } else {
            R.resize(world);
            for (int r = 0; r < world; ++r) {
                R[r].rank = r; R[r].device = c.devices[r];
                for (int q = 0; q < r; ++q) shared_device |= c.devices[q] == c.devices[r];
            }
        }
        for (auto& r : R) if (r.device < 0 || r.device >= ndev) throw EngineError(SPHMI_ERR_ARGUMENT, "device ordinal out of range");
        const char* tr = getenv("SPHMI_TRANSPORT");
        // (a one-rank world under sphmi_create_rank still runs its allreduce through RCCL: the binding is exercised)
        const bool want_shm = tr && !strcmp(tr, "shm");
        if (want_shm && !rank_mode) throw EngineError(SPHMI_ERR_ARGUMENT, "SPHMI_TRANSPORT=shm is the transport of sphmi_create_rank (one slab per process)");
        // RCCL itself refuses two ranks on one device; a SUBSTITUTE library ($SPHMI_RCCL_LIB: the checking double of tests/mock_rccl/) may not,
        // and SPHMI_TRANSPORT=rccl then sends the slabs of a one-process handle that share a device through the ncclCommInitAll branch below
        const char* sub = getenv("SPHMI_RCCL_LIB");
        const bool want_rccl = tr && !strcmp(tr, "rccl"), substitute = sub && *sub;
        use_rccl = (rank_mode && !want_shm) || (!rank_mode && world > 1 && !(tr && !strcmp(tr, "local")) && (!shared_device || (want_rccl && substitute)));
        if (want_rccl && shared_device && !substitute) throw EngineError(SPHMI_ERR_ARGUMENT, "RCCL cannot run two ranks on one device");
        if (want_shm) {
            if (!unique_id) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_create_rank: null unique id");
            shm.reset(new ShmWorld(unique_id, 128, my_rank, world));
            HC(hipHostMalloc((void**)&red_h, 4 * 8));
        }
        if (use_rccl) {
            Rccl& N = Rccl::get();
            const char* oc = getenv("SPHMI_RCCL_ONE_COMM");
            const bool one_comm = oc && atoi(oc) != 0;
            if (rank_mode) {
                if (!unique_id) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_create_rank: null unique id");
                ncclUniqueId id; memcpy(&id, unique_id, sizeof id);
                HC(hipSetDevice(R[0].device));
                NCX("ncclCommInitRank (point-to-point communicator)", my_rank, -1, nullptr, N.CommInitRank(&R[0].comm, world, id, my_rank));
                R[0].comm_red = R[0].comm;
                if (world > 1 && !one_comm) {
                    // the id of the second communicator: made on rank 0, summed over the first one (the others add zeros)
                    ncclUniqueId id2; memset(&id2, 0, sizeof id2);
                    if (my_rank == 0) NCX("ncclGetUniqueId (allreduce communicator)", my_rank, -1, nullptr, N.GetUniqueId(&id2));
                    void* d = nullptr;
                    HC(hipMalloc(&d, sizeof id2));
                    try {
                        HostBounce hb;                       // (the slab engines do not exist yet)
                        hb.h2d(d, &id2, sizeof id2, nullptr);
                        NCX("ncclAllReduce (handing the second communicator's id round)", my_rank, -1, R[0].comm, N.AllReduce(d, d, sizeof id2, ncclUint8, ncclSum, R[0].comm, nullptr));
                        HC(hipStreamSynchronize(nullptr));
                        hb.d2h(&id2, d, sizeof id2, nullptr);
                    } catch (...) { (void)hipFree(d); throw; }
                    (void)hipFree(d);
                    NCX("ncclCommInitRank (allreduce communicator)", my_rank, -1, nullptr, N.CommInitRank(&R[0].comm_red, world, id2, my_rank));
                }
            } else {
                std::vector<ncclComm_t> comms(world);
                std::vector<int> devs(world);
                for (int r = 0; r < world; ++r) devs[r] = R[r].device;
                NCX("ncclCommInitAll (point-to-point communicators)", 0, -1, nullptr, N.CommInitAll(comms.data(), world, devs.data()));
                for (int r = 0; r < world; ++r) R[r].comm = R[r].comm_red = comms[r];
                if (!one_comm) {
                    NCX("ncclCommInitAll (allreduce communicators)", 0, -1, nullptr, N.CommInitAll(comms.data(), world, devs.data()));
                    for (int r = 0; r < world; ++r) R[r].comm_red = comms[r];
                }
            }
        }
        if (!use_rccl && !rank_mode) {
            // stream-ordered copies between the slabs of this process: peers on other devices need peer access
            for (auto& a : R) for (auto& b : R) if (a.device != b.device) {
                HC(hipSetDevice(a.device));
                int can = 0; (void)hipDeviceCanAccessPeer(&can, a.device, b.device);
                if (can) { hipError_t e = hipDeviceEnablePeerAccess(b.device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { can = 0; } (void)hipGetLastError(); }
                if (!can) peer_ok = false;
            }
        }
        for (auto& r : R) {
            r.has_left = r.rank > 0; r.has_right = r.rank < world - 1;
            HC(hipSetDevice(r.device));
            {   // $SPHMI_SIDE_PRIORITY=1: the exchange chain (halo -> unpack -> slab-edge tiles) is dispatched ahead of the interior tiles it runs beside
                const char* sp = getenv("SPHMI_SIDE_PRIORITY");
                int lo = 0, hi = 0;
                if (sp && atoi(sp) != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) HC(hipStreamCreateWithPriority(&r.side, hipStreamNonBlocking, hi));
                else HC(hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking));
            }
            HC(hipEventCreateWithFlags(&r.ev_pack, hipEventDisableTiming));
            HC(hipEventCreateWithFlags(&r.ev_edge, hipEventDisableTiming));
            HC(hipEventCreateWithFlags(&r.ev_red, hipEventDisableTiming));
            for (auto& e : r.ev_consumed) HC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HC(hipMalloc(&r.M, 4 * 8)); HC(hipMalloc(&r.stage, 16 * 4 * 8));
            HC(hipMemset(r.M, 0, 4 * 8));
            HC(hipMalloc(&r.mm_d, 2 * 4)); HC(hipHostMalloc(&r.mm_h, 2 * 4)); HC(hipHostMalloc(&r.cnt_h, 8 * 8));
        }
        if (const char* x = getenv("SPHMI_EXCHANGE")) {
            if (!strcmp(x, "mailbox")) setup_mailboxes();
            else if (strcmp(x, "allreduce") != 0) throw EngineError(SPHMI_ERR_ARGUMENT, "SPHMI_EXCHANGE must be allreduce (default) or mailbox");
        }
    }
    // $SPHMI_EXCHANGE=mailbox (k_dd_mbox_merge_control): the per-step maxima through mailboxes in device memory instead of ncclAllReduce / the host
    // round trip of the shared-memory transport.  The slabs of ONE process that talk through device pointers anyway (the local transport) have no use for it.
    void setup_mailboxes() {
        if (world < 2 || (!use_rccl && !shm)) return;
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            // FINE-GRAINED device memory: a peer device (or process) writes the box over xGMI / hipIpc WHILE this slab's control kernel polls it, and
            // coarse-grained allocations are only guaranteed coherent across devices at kernel boundaries — a poll served from the local L2 could
            // stay stale until the deadline (round-5 advisor finding)
            HC(hipExtMallocWithFlags((void**)&r.box, kMboxWords * 8, hipDeviceMallocFinegrained));
            HC(hipMemset(r.box, 0, kMboxWords * 8));
            HC(hipDeviceSynchronize());
        }
        if (rank_mode) {
            // one process per slab: the boxes of the peers through hipIpc — the 64-byte handles summed into one table (everybody adds zeros but its own row)
            Rank& r = R[0];
            HC(hipSetDevice(r.device));
            static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t: 64 bytes per row");
            std::vector<long long> tab((size_t)world * 8, 0);
            hipIpcMemHandle_t mine_h;
            HC(hipIpcGetMemHandle(&mine_h, r.box));
            memcpy(&tab[(size_t)r.rank * 8], &mine_h, 64);
            if (shm) {
                // (a sum of int64 words: every row but one is zero, so nothing carries)
                shm->allreduce((int64_t*)tab.data(), tab.size(), ShmWorld::SUM);
            } else {
                void* d = nullptr;
                HC(hipMalloc(&d, tab.size() * 8));
                try {
                    HostBounce hb;
                    hb.h2d(d, tab.data(), tab.size() * 8, nullptr);
                    NCX("ncclAllReduce (handing the mailbox handles round)", r.rank, -1, r.comm_red, Rccl::get().AllReduce(d, d, tab.size() * 8, ncclUint8, ncclSum, r.comm_red, nullptr));
                    HC(hipStreamSynchronize(nullptr));
                    hb.d2h(tab.data(), d, tab.size() * 8, nullptr);
                } catch (...) { (void)hipFree(d); throw; }
                (void)hipFree(d);
            }
            for (int q = 0; q < world; ++q) {
                if (q == r.rank) { r.box_of[q] = r.box; continue; }
                hipIpcMemHandle_t hq; memcpy(&hq, &tab[(size_t)q * 8], 64);
                void* pq = nullptr;
                if (hipIpcOpenMemHandle(&pq, hq, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !pq) {
                    (void)hipGetLastError();
                    throw EngineError(SPHMI_ERR_DEVICE, "SPHMI_EXCHANGE=mailbox: hipIpcOpenMemHandle failed for the mailbox of slab " + std::to_string(q));
                }
                r.box_ipc[q] = pq; r.box_of[q] = (unsigned long long*)pq;
            }
        } else {
            // the slabs of this process on their own devices: peer access, plain pointers
            for (auto& a : R) for (auto& b : R) {
                if (a.device != b.device) {
                    HC(hipSetDevice(a.device));
                    int can = 0; (void)hipDeviceCanAccessPeer(&can, a.device, b.device);
                    if (can) { hipError_t e = hipDeviceEnablePeerAccess(b.device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0; (void)hipGetLastError(); }
                    if (!can) throw EngineError(SPHMI_ERR_DEVICE, "SPHMI_EXCHANGE=mailbox needs peer access between the devices of the handle");
                }
                a.box_of[b.rank] = b.box;
            }
        }
        double secs = 20.0;
        if (const char* t = getenv("SPHMI_MBOX_TIMEOUT")) secs = atof(t) > 0 ? atof(t) : secs;
        mbox_timeout = (unsigned long long)(secs * 1e8);            // s_memrealtime: 100 MHz
        mbox_on = true;
    }
    ~MultiEngine() override {
        for (auto& r : R) {
            (void)hipSetDevice(r.device);
            if (r.e) (void)hipStreamSynchronize(r.e->stream);
            if (r.side) { (void)hipStreamSynchronize(r.side); }
            if (r.comm_red && r.comm_red != r.comm) { try { Rccl::get().CommDestroy(r.comm_red); } catch (...) {} }
            if (r.comm) { try { Rccl::get().CommDestroy(r.comm); } catch (...) {} }
            for (auto& h : r.halo) for (DevBuf* b : {&h.send_l, &h.send_r, &h.slot_l, &h.slot_r, &h.sb_l, &h.sb_r, &h.rb_l, &h.rb_r}) b->release();
            stage_s.release(); stage_r.release();
            if (red_h) { (void)hipHostFree(red_h); red_h = nullptr; }
            for (DevBuf* b : {&r.cx, &r.flag, &r.pos, &r.idx[0], &r.idx[1], &r.idx[2], &r.idx[3], &r.rec_s[0], &r.rec_s[1], &r.rec_r[0], &r.rec_r[1], &r.cost}) b->release();
            for (void* q : r.box_ipc) if (q) (void)hipIpcCloseMemHandle(q);
            if (r.box) (void)hipFree(r.box);
            (void)hipFree(r.M); (void)hipFree(r.stage); (void)hipFree(r.mm_d); (void)hipHostFree(r.mm_h); (void)hipHostFree(r.cnt_h);
            if (r.ev_pack) { (void)hipEventDestroy(r.ev_pack); (void)hipEventDestroy(r.ev_edge); (void)hipEventDestroy(r.ev_red); }
            for (auto& e : r.ev_consumed) if (e) (void)hipEventDestroy(e);
            r.e.reset();
            if (r.side) (void)hipStreamDestroy(r.side);
        }
    }

    // ---- host-side collectives (rebuild-time bookkeeping: counts and the handful of scalars of the re-cut) --------
    // vals[k][r]: value k of local rank r → every local rank sees the reduction over ALL ranks
    enum Op { OP_SUM, OP_MAX };
    void host_allreduce(std::vector<long long>& v /* per local rank: n values each, concatenated [r][k] */, int n, Op op) {
        const int L = (int)R.size();
        if (!rank_mode) {
            for (int k = 0; k < n; ++k) {
                long long acc = v[k];
                for (int r = 1; r < L; ++r) acc = op == OP_SUM ? acc + v[(size_t)r * n + k] : std::max(acc, v[(size_t)r * n + k]);
                for (int r = 0; r < L; ++r) v[(size_t)r * n + k] = acc;
            }
            return;
        }
        if (world == 1) return;
        if (shm) { shm->allreduce((int64_t*)v.data(), (size_t)n, op == OP_SUM ? ShmWorld::SUM : ShmWorld::MAX); return; }
        Rank& r = R[0];
        HC(hipSetDevice(r.device));
        void* d = r.cost.need((size_t)n * 8);
        r.e->bounce.h2d(d, v.data(), (size_t)n * 8, r.main);
        NCX("ncclAllReduce (rebuild-time host scalars)", r.rank, -1, r.comm_red, Rccl::get().AllReduce(d, d, (size_t)n, ncclInt64, op == OP_SUM ? ncclSum : ncclMax, r.comm_red, r.main));
        r.e->bounce.d2h(v.data(), d, (size_t)n * 8, r.main);
    }
    // every local rank tells its neighbours one number per side; returns what the neighbours told it
    void host_neighbour_counts(const std::vector<long long>& to_l, const std::vector<long long>& to_r,
                               std::vector<long long>& from_l, std::vector<long long>& from_r) {
        const int L = (int)R.size();
        from_l.assign(L, 0); from_r.assign(L, 0);
        if (!rank_mode) {
            for (int r = 0; r < L; ++r) { if (r > 0) from_l[r] = to_r[r - 1]; if (r + 1 < L) from_r[r] = to_l[r + 1]; }
            return;
        }
        if (world == 1) return;
        Rank& r = R[0];
        if (shm) {
            long long out[2] = {to_l[0], to_r[0]}, in[2] = {0, 0};
            std::vector<ShmWorld::Xfer> x;
            if (r.has_left) { x.push_back({true, 0, (char*)&out[0], 8, 0}); x.push_back({false, 0, (char*)&in[0], 8, 0}); }
            if (r.has_right) { x.push_back({true, 1, (char*)&out[1], 8, 0}); x.push_back({false, 1, (char*)&in[1], 8, 0}); }
            shm->exchange(x);
            from_l[0] = in[0]; from_r[0] = in[1];
            return;
        }
        HC(hipSetDevice(r.device));
        long long* d = (long long*)r.cost.need(4 * 8);
        r.cnt_h[0] = to_l[0]; r.cnt_h[1] = to_r[0]; r.cnt_h[2] = 0; r.cnt_h[3] = 0;
        HC(hipMemcpyAsync(d, r.cnt_h, 4 * 8, hipMemcpyHostToDevice, r.main));
        Rccl& N = Rccl::get();
        NCX("ncclGroupStart (neighbour counts)", r.rank, -1, r.comm, N.GroupStart());
        if (r.has_left) { NCX("ncclSend (neighbour counts)", r.rank, r.rank - 1, r.comm, N.Send(d + 0, 1, ncclInt64, r.rank - 1, r.comm, r.main)); NCX("ncclRecv (neighbour counts)", r.rank, r.rank - 1, r.comm, N.Recv(d + 2, 1, ncclInt64, r.rank - 1, r.comm, r.main)); }
        if (r.has_right) { NCX("ncclSend (neighbour counts)", r.rank, r.rank + 1, r.comm, N.Send(d + 1, 1, ncclInt64, r.rank + 1, r.comm, r.main)); NCX("ncclRecv (neighbour counts)", r.rank, r.rank + 1, r.comm, N.Recv(d + 3, 1, ncclInt64, r.rank + 1, r.comm, r.main)); }
        NCX("ncclGroupEnd (neighbour counts)", r.rank, -1, r.comm, N.GroupEnd());
        HC(hipMemcpyAsync(r.cnt_h, d, 4 * 8, hipMemcpyDeviceToHost, r.main));
        HC(hipStreamSynchronize(r.main));
        from_l[0] = r.cnt_h[2]; from_r[0] = r.cnt_h[3];
    }

    // ---- device point-to-point phase ----------------------------------------------------------------------------
    // Every local sender has recorded ev_pack on its main stream after filling its send buffers.  side: the messages
    // (and whatever the caller queues behind them) go to the side streams, so that the interior launch on main overlaps.
    void exchange(const std::vector<Msg>& msgs, bool side) {
        if (msgs.empty()) return;
        if (shm) {
            // device → page-locked host → shared-memory ring → peer's host → peer's device, on the stream the phase runs on
            Rank& r = R[0];
            hipStream_t q = side ? r.side : r.main;
            HC(hipSetDevice(r.device));
            size_t tot_s = 0, tot_r = 0;
            for (const Msg& m : msgs) { if (m.src == r.rank) tot_s += m.bytes; if (m.dst == r.rank) tot_r += m.bytes; }
            char* hs = stage_s.need(tot_s + 1); char* hr = stage_r.need(tot_r + 1);
            size_t os = 0, orr = 0;
            for (const Msg& m : msgs) if (m.src == r.rank && m.bytes) { HC(hipMemcpyAsync(hs + os, m.sbuf, m.bytes, hipMemcpyDeviceToHost, q)); os += m.bytes; }
            HC(hipStreamSynchronize(q));
            std::vector<ShmWorld::Xfer> x;
            os = 0;
            for (const Msg& m : msgs) {
                if (m.src == r.rank) { x.push_back({true, m.dst > r.rank ? 1 : 0, hs + os, m.bytes, 0}); os += m.bytes; }
                if (m.dst == r.rank) { x.push_back({false, m.src > r.rank ? 1 : 0, hr + orr, m.bytes, 0}); orr += m.bytes; }
            }
            shm->exchange(x);
            orr = 0;
            for (const Msg& m : msgs) if (m.dst == r.rank && m.bytes) { HC(hipMemcpyAsync(m.rbuf, hr + orr, m.bytes, hipMemcpyHostToDevice, q)); orr += m.bytes; }
            HC(hipStreamSynchronize(q));                                  // the staging buffers serve the next phase
            return;
        }
        if (use_rccl) {
            Rccl& N = Rccl::get();
            const char* what = side ? "halo / records, side stream" : "halo / records, main stream";
            NCX(what, R[0].rank, -1, R[0].comm, N.GroupStart());
            for (const Msg& m : msgs) {
                if (Rank* s = local(m.src)) { HC(hipSetDevice(s->device)); NCX(what, s->rank, m.dst, s->comm, N.Send(m.sbuf, m.bytes, ncclUint8, m.dst, s->comm, side ? s->side : s->main)); }
                if (Rank* d = local(m.dst)) { HC(hipSetDevice(d->device)); NCX(what, d->rank, m.src, d->comm, N.Recv(m.rbuf, m.bytes, ncclUint8, m.src, d->comm, side ? d->side : d->main)); }
            }
            NCX(what, R[0].rank, -1, R[0].comm, N.GroupEnd());
            return;
        }
        for (const Msg& m : msgs) {
            Rank *s = local(m.src), *d = local(m.dst);
            hipStream_t q = side ? d->side : d->main;
            HC(hipSetDevice(d->device));
            HC(hipStreamWaitEvent(q, s->ev_pack, 0));
            if (m.bytes) {
                // (sbuf / rbuf are 16-byte aligned device allocations; across devices the kernel needs peer access — else the runtime's copy)
                if (s->device == d->device || peer_ok) {
                    hipLaunchKernelGGL(k_dd_copy, dim3((unsigned)std::min<size_t>((m.bytes / 16 + 255) / 256 + 1, 2048)), dim3(256), 0, q, m.sbuf, m.rbuf, m.bytes);
                    HC(hipGetLastError());
                } else HC(hipMemcpyAsync(m.rbuf, m.sbuf, m.bytes, hipMemcpyDeviceToDevice, q));
            }
            if (m.slot >= 0) {       // the sender must not refill this buffer before the copy has read it
                HC(hipEventRecord(s->ev_consumed[m.slot], q));
                s->consumed_valid[m.slot] = true;
            }
        }
    }
    void wait_consumed(Rank& r, int slot) {
        if (!use_rccl && !shm && r.consumed_valid[slot]) { HC(hipStreamWaitEvent(r.main, r.ev_consumed[slot], 0)); r.consumed_valid[slot] = false; }
    }

    // ---- upload: split the particle set, one slab engine per local rank ------------------------------------------
    void upload(const void* position, const void* velocity, const void* acceleration, const void* density,
                const uint8_t* ty, const int64_t* ids, const uint64_t* groups, const void* ghost_points) override {
        if (!position || !velocity || !density || !ty || !ids) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: null array");
        if (cfg.mdbc != SPHMI_MDBC_NONE && !ghost_points) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: mDBC handle without ghost points");
        const int64_t N = cfg.n_particles;
        n_total = N;
        SlabSetup S;
        plan_slabs(cfg, position, ghost_points, N, world, cfg.slab_axis - 1, given_plan.world() == world ? &given_plan : nullptr, 1.6, S);
        axis = S.axis; halo_width = S.halo_width; plan = S.plan;
        const size_t hb = (size_t)cfg.host_float_bytes;
        // the whole set is checked, whichever slabs are local (Engine::upload's checks see the local slabs only)
        for (int64_t i = 0; i < N; ++i) {
            if (ty[i] < 1 || ty[i] > 3) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: ParticleType must be 1, 2 or 3");
            const double rho = hb == 8 ? ((const double*)density)[i] : (double)((const float*)density)[i];
            if (!(rho > 0.0)) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_upload: density must be positive");
        }
        // ONE pass over the owner column: the index lists of the LOCAL slabs (counting sort by owner; a rank-mode process
        // keeps one list — the other slabs' particles are never copied)
        std::vector<int> local_of(world, -1);
        for (size_t q = 0; q < R.size(); ++q) local_of[R[q].rank] = (int)q;
        std::vector<size_t> n_of(R.size(), 0);
        for (int64_t i = 0; i < N; ++i) { const int q = local_of[S.owner[(size_t)i]]; if (q >= 0) n_of[q] += 1; }
        std::vector<std::vector<int64_t>> mine_of(R.size());
        for (size_t q = 0; q < R.size(); ++q) mine_of[q].reserve(n_of[q]);
        for (int64_t i = 0; i < N; ++i) { const int q = local_of[S.owner[(size_t)i]]; if (q >= 0) mine_of[q].push_back(i); }
        for (size_t q_ = 0; q_ < R.size(); ++q_) {
            Rank& r = R[q_];
            HC(hipSetDevice(r.device));
            sphmi_config c = cfg;
            c.n_particles = S.capacity[r.rank]; c.device = r.device; c.n_devices = 0;
            if (c.n_particles >= (cfg.device_float_bytes == 8 ? (1ll << 26) : (1ll << 27))) throw EngineError(SPHMI_ERR_ARGUMENT, "slab capacity beyond 2^27 particles: use more devices");
            r.e.reset(new Engine<T>(c));
            r.main = r.e->stream;
            r.e->dd_set_slab(axis, std::max(plan.lo[r.rank], -SlabPlan::INF), std::min(plan.hi[r.rank], SlabPlan::INF), r.has_left, r.has_right);
            for (int m = 0; m < motions_n; ++m) r.e->set_motion(mot_group[m], mot_vel[m], mot_start[m], mot_dur[m], mot_dir[m]);
            r.e->iteration = iteration; r.e->total_time = total_time;
            const std::vector<int64_t>& mine = mine_of[q_];
            const size_t n = mine.size();
            if (n == 0) throw EngineError(SPHMI_ERR_ARGUMENT, "a slab owns no particle: too many devices for this case");
            auto take = [&](const void* src, size_t elem) {
                std::vector<char> o(src ? n * elem : 0);
                if (src) for (size_t k = 0; k < n; ++k) memcpy(&o[k * elem], (const char*)src + (size_t)mine[k] * elem, elem);
                return o;
            };
            auto px = take(position, hb * D), pv = take(velocity, hb * D), pa = take(acceleration, hb * D), pr = take(density, hb);
            auto pt = take(ty, 1), pi = take(ids, 8), pg = take(groups, 8), ph = take(ghost_points, hb * D);
            r.e->dd_upload((int64_t)n, px.data(), pv.data(), acceleration ? pa.data() : nullptr, pr.data(), (const uint8_t*)pt.data(),
                           (const int64_t*)pi.data(), groups ? (const uint64_t*)pg.data() : nullptr,
                           ghost_points ? ph.data() : nullptr, mine.data());
        }
        uploaded = true; have_halo = false; recut_ready = false; dx_rate = 0.0; parity = 0;
    }

    SlabPlan given_plan;               // test hook (sphmi_multi_set_cuts): start from these cuts instead of the balanced ones
    int motions_n = 0; uint64_t mot_group[16]; double mot_vel[16], mot_start[16], mot_dur[16], mot_dir[16][3];
    void set_motion(uint64_t group, double vel, double start, double dur, const double* dir) override {
        if (!dir) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_motion: null direction");
        int m = 0;
        while (m < motions_n && mot_group[m] != group) ++m;
        if (m == 16) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_motion: more than 16 moving groups");
        if (m == motions_n) motions_n += 1;
        mot_group[m] = group; mot_vel[m] = vel; mot_start[m] = start; mot_dur[m] = dur;
        for (int d = 0; d < 3; ++d) mot_dir[m][d] = d < D ? dir[d] : 0.0;
        moving = true;
        for (auto& r : R) if (r.e) r.e->set_motion(group, vel, start, dur, dir);
    }

    // ---- collective rebuild ------------------------------------------------------------------------------------
    int* where(Rank& r, int tmask, int tval, bool any_live, long long lo, long long hi, DevBuf& out, int& n_out, bool sync_count = true) {
        Engine<T>& e = *r.e;
        const int N = e.N;
        n_out = 0;
        if (N == 0) return nullptr;
        int* flag = (int*)r.flag.need((size_t)N * 4);
        int* pos = (int*)r.pos.need((size_t)(N + 1) * 4);
        const int nb = (N + 255) / 256;
        hipLaunchKernelGGL(k_dd_flag, dim3(nb), dim3(256), 0, r.main, (const int*)r.cx.p, (const uint8_t*)e.type[e.cur], N, tmask, tval, any_live ? 1 : 0, lo, hi, flag);
        const int ntiles = (N + kScanTile - 1) / kScanTile;
        int* tsum = (int*)r.cost.need((size_t)(ntiles + 8) * 4 + 64);
        hipLaunchKernelGGL(k_scan_tile, dim3(ntiles), dim3(kScanThreads), 0, r.main, (const int*)flag, pos, N, tsum, tsum + ntiles + 2);
        hipLaunchKernelGGL(k_scan_tsums, dim3(1), dim3(1024), 0, r.main, tsum, ntiles, tsum + ntiles + 1);
        hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(kScanThreads), 0, r.main, pos, N, (const int*)tsum, (const int*)(tsum + ntiles + 1));
        HC(hipGetLastError());
        HC(hipMemcpyAsync(r.mm_h, pos + N, 4, hipMemcpyDeviceToHost, r.main));
        HC(hipStreamSynchronize(r.main));
        n_out = r.mm_h[0];
        int* idx = (int*)out.need((size_t)std::max(n_out, 1) * 4);
        if (n_out) { hipLaunchKernelGGL(k_dd_compact, dim3(nb), dim3(256), 0, r.main, (const int*)flag, (const int*)pos, N, idx); HC(hipGetLastError()); }
        return idx;
    }
    void cellx(Rank& r) {
        Engine<T>& e = *r.e;
        int* cx = (int*)r.cx.need((size_t)std::max(e.N, 1) * 4);
        e.dd_cell_x_dev(cx);
    }
    void set_slab(Rank& r) {
        r.e->dd_set_slab(axis, std::max(plan.lo[r.rank], -SlabPlan::INF), std::min(plan.hi[r.rank], SlabPlan::INF), r.has_left, r.has_right);
    }

    void rebuild_collective() {
        const int L = (int)R.size();
        for (auto& r : R) { HC(hipSetDevice(r.device)); cellx(r); }
        // 0. load balance by WORK (candidates per particle on the cell list of the previous rebuild): the rebuild is the
        //    only time particles change cells, so it is also when the cuts may move
        // (`recut_ready`: the slab engines hold a cell list — false until the first rebuild after an upload, which makes NEW slab engines; until round 6 the
        // test was n_rebuilds > 0, and a second sphmi_upload on a multi-device handle ended in "sphmi_dd_column_cost before the first rebuild")
        if (world > 1 && recut_imbalance > 0 && recut_ready) {
            std::vector<long long> ext((size_t)L * 2);
            for (int q = 0; q < L; ++q) {
                Rank& r = R[q]; HC(hipSetDevice(r.device));
                r.mm_h[0] = INT32_MAX; r.mm_h[1] = INT32_MIN;
                HC(hipMemcpyAsync(r.mm_d, r.mm_h, 8, hipMemcpyHostToDevice, r.main));
                hipLaunchKernelGGL(k_dd_minmax, dim3(std::min((r.e->N + 255) / 256, 512)), dim3(256), 0, r.main, (const int*)r.cx.p, (const uint8_t*)r.e->type[r.e->cur], (const int*)nullptr, r.e->N, r.mm_d);
                HC(hipMemcpyAsync(r.mm_h, r.mm_d, 8, hipMemcpyDeviceToHost, r.main));
                HC(hipStreamSynchronize(r.main));
                ext[(size_t)q * 2] = -(long long)r.mm_h[0]; ext[(size_t)q * 2 + 1] = r.mm_h[1];
            }
            host_allreduce(ext, 2, OP_MAX);
            const long long gmin = -ext[0], gmax = ext[1];
            const int ncols = (int)(gmax - gmin + 1);
            std::vector<std::vector<long long>> cost(L, std::vector<long long>(ncols, 0));
            std::vector<long long> tot((size_t)L * 2);
            for (int q = 0; q < L; ++q) {
                Rank& r = R[q]; HC(hipSetDevice(r.device));
                unsigned long long* cd = (unsigned long long*)r.cost.need((size_t)ncols * 8);
                HC(hipMemsetAsync(cd, 0, (size_t)ncols * 8, r.main));
                r.e->dd_column_cost(gmin, ncols, (uint64_t*)cd);
                r.e->bounce.d2h(cost[q].data(), cd, (size_t)ncols * 8, r.main);
                long long mine = 0; for (auto v : cost[q]) mine += v;
                tot[(size_t)q * 2] = mine;
            }
            // heaviest rank and total
            std::vector<long long> mx(L), sm(L);
            for (int q = 0; q < L; ++q) { mx[q] = tot[(size_t)q * 2]; sm[q] = tot[(size_t)q * 2]; }
            host_allreduce(mx, 1, OP_MAX); host_allreduce(sm, 1, OP_SUM);
            if ((double)mx[0] * world > recut_imbalance * (double)sm[0]) {
                std::vector<long long> hist((size_t)L * ncols);
                for (int q = 0; q < L; ++q) memcpy(&hist[(size_t)q * ncols], cost[q].data(), (size_t)ncols * 8);
                host_allreduce(hist, ncols, OP_SUM);
                std::vector<double> h(ncols);
                for (int k = 0; k < ncols; ++k) h[k] = (double)hist[k];
                SlabPlan np = plan_recut(plan, gmin, h);
                if (np.cuts() != plan.cuts()) { plan = np; for (auto& r : R) set_slab(r); n_recuts += 1; }
            }
        }
        // 1. ghosts die, leavers migrate to the adjacent rank
        std::vector<long long> to_l(L, 0), to_r(L, 0), from_l, from_r;
        std::vector<int*> go_l(L, nullptr), go_r(L, nullptr);
        for (int q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            const long long lo = plan.lo[r.rank], hi = plan.hi[r.rank];
            int nl = 0, nr = 0;
            if (r.has_left) go_l[q] = where(r, kGhostMask, 0, false, -(1ll << 40), lo - 1, r.idx[0], nl);
            if (r.has_right) go_r[q] = where(r, kGhostMask, 0, false, hi + 1, 1ll << 40, r.idx[1], nr);
            to_l[q] = nl; to_r[q] = nr;
            // a particle must not skip a whole slab between two rebuilds
            for (int s = 0; s < 2; ++s) {
                const int n = s == 0 ? nl : nr;
                if (!n) continue;
                r.mm_h[0] = INT32_MAX; r.mm_h[1] = INT32_MIN;
                HC(hipMemcpyAsync(r.mm_d, r.mm_h, 8, hipMemcpyHostToDevice, r.main));
                hipLaunchKernelGGL(k_dd_minmax, dim3(std::min((n + 255) / 256, 512)), dim3(256), 0, r.main, (const int*)r.cx.p, (const uint8_t*)r.e->type[r.e->cur], (const int*)(s == 0 ? go_l[q] : go_r[q]), n, r.mm_d);
                HC(hipMemcpyAsync(r.mm_h, r.mm_d, 8, hipMemcpyDeviceToHost, r.main));
                HC(hipStreamSynchronize(r.main));
                if ((s == 0 && r.mm_h[0] < plan.lo[r.rank - 1]) || (s == 1 && r.mm_h[1] > plan.hi[r.rank + 1]))
                    throw EngineError(SPHMI_ERR_DOMAIN, "domain decomposition: a particle skipped a whole slab between two rebuilds");
            }
            r.e->dd_kill_ghosts();
        }
        migrate(to_l, to_r, go_l, go_r, /*kill=*/true, 0, 0);
        // (rounds 2-4 sorted here — dead rows out, arrivals into cell order — and again behind the ghost layers.  Round 5: ONE sort.  Nothing below needs the
        // order: the boundary columns are found by cell column on the unsorted rows (dead rows have type 0), and the in-cell order of the final sort is the
        // order TAGS' on every slab — owned rows, arrivals and ghost copies alike carry the tags of the previous global order — so sender lists and ghost
        // slots pair up exactly as after two sorts.  $SPHMI_DD_TWO_SORTS=1: the old sequence.)
        if (two_sorts) for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); r.e->dd_rebuild(); }
        // 2. the first / last column(s) of the slab become the neighbours' ghost layer
        const int W = halo_width;
        std::vector<long long> n_bl(L, 0), n_br(L, 0);
        for (int q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            cellx(r);
            const long long lo = plan.lo[r.rank], hi = plan.hi[r.rank];
            int nl = 0, nr = 0;
            if (r.has_left) go_l[q] = where(r, 0, 0, true, -(1ll << 40), lo + W - 1, r.idx[0], nl);
            if (r.has_right) go_r[q] = where(r, 0, 0, true, hi - W + 1, 1ll << 40, r.idx[1], nr);
            n_bl[q] = nl; n_br[q] = nr;
        }
        std::vector<long long> got_l, got_r;
        migrate(n_bl, n_br, go_l, go_r, /*kill=*/false, kGhostLeft, kGhostRight, &got_l, &got_r);
        for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); r.e->dd_rebuild(); }
        // 3. halo index lists in the final order (stable sorts ⇒ k-th sender entry ↔ k-th ghost slot; the same holds for
        //    the sub-lists of ONE column).  State A travels with all halo_width columns (mDBC reads them), the
        //    half-step state H with the one column the pair forces reach.
        for (int q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            cellx(r);
            const long long lo = plan.lo[r.rank], hi = plan.hi[r.rank];
            for (int k = 0; k < 2; ++k) {
                const int w = k == 0 ? W : 1;
                Halo& h = r.halo[k];
                h.n_send_l = h.n_send_r = h.n_slot_l = h.n_slot_r = 0;
                if (r.has_left) where(r, kGhostMask, 0, false, -(1ll << 40), lo + w - 1, h.send_l, h.n_send_l);
                if (r.has_right) where(r, kGhostMask, 0, false, hi - w + 1, 1ll << 40, h.send_r, h.n_send_r);
                // (a slab without a neighbour on a side holds no ghost copies from it: no list, no round trip for its length)
                if (r.has_left) where(r, kGhostLeft, kGhostLeft, false, lo - w, 1ll << 40, h.slot_l, h.n_slot_l);
                if (r.has_right) where(r, kGhostRight, kGhostRight, false, -(1ll << 40), hi + w, h.slot_r, h.n_slot_r);
                if (k == 0 && (h.n_send_l != n_bl[q] || h.n_send_r != n_br[q] || h.n_slot_l != got_l[q] || h.n_slot_r != got_r[q]))
                    throw EngineError(SPHMI_ERR_STATE, "domain decomposition: boundary columns changed between the two sorts");
                const size_t vb = 2 * sizeof(V4);
                h.sb_l.need((size_t)std::max(h.n_send_l, 1) * vb); h.sb_r.need((size_t)std::max(h.n_send_r, 1) * vb);
                h.rb_l.need((size_t)std::max(h.n_slot_l, 1) * vb); h.rb_r.need((size_t)std::max(h.n_slot_r, 1) * vb);
            }
            for (bool& v : r.consumed_valid) v = false;
        }
        // the one-column lists of state H must pair up across the cut exactly like the wide ones
        {
            std::vector<long long> sl(L), sr(L), fl, fr;
            for (int q = 0; q < L; ++q) { sl[q] = R[q].halo[1].n_send_l; sr[q] = R[q].halo[1].n_send_r; }
            host_neighbour_counts(sl, sr, fl, fr);
            for (int q = 0; q < L; ++q)
                if (R[q].halo[1].n_slot_l != fl[q] || R[q].halo[1].n_slot_r != fr[q])
                    throw EngineError(SPHMI_ERR_STATE, "domain decomposition: edge-column lists do not pair up across a cut");
        }
        for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); }
        have_halo = true;
        recut_ready = true;
        n_rebuilds += 1;
    }
    bool recut_ready = false;

    // full records of the listed particles to the adjacent ranks; arrivals are appended with `flag`
    void migrate(const std::vector<long long>& n_l, const std::vector<long long>& n_r, const std::vector<int*>& idx_l,
                 const std::vector<int*>& idx_r, bool kill, int flag_from_left, int flag_from_right,
                 std::vector<long long>* got_l = nullptr, std::vector<long long>* got_r = nullptr) {
        const int L = (int)R.size();
        std::vector<long long> from_l, from_r;
        host_neighbour_counts(n_l, n_r, from_l, from_r);
        std::vector<Msg> msgs;
        for (int q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            void* sl = n_l[q] ? r.rec_s[0].need(DdRecord<T>::bytes((size_t)n_l[q])) : nullptr;
            void* sr = n_r[q] ? r.rec_s[1].need(DdRecord<T>::bytes((size_t)n_r[q])) : nullptr;
            if (n_l[q]) e.dd_gather(idx_l[q], n_l[q], sl);
            if (n_r[q]) e.dd_gather(idx_r[q], n_r[q], sr);
            void* rl = from_l[q] ? r.rec_r[0].need(DdRecord<T>::bytes((size_t)from_l[q])) : nullptr;
            void* rr = from_r[q] ? r.rec_r[1].need(DdRecord<T>::bytes((size_t)from_r[q])) : nullptr;
            HC(hipEventRecord(r.ev_pack, r.main));
            if (n_l[q]) msgs.push_back({r.rank, r.rank - 1, sl, nullptr, DdRecord<T>::bytes((size_t)n_l[q]), -1});
            if (n_r[q]) msgs.push_back({r.rank, r.rank + 1, sr, nullptr, DdRecord<T>::bytes((size_t)n_r[q]), -1});
            if (from_l[q]) msgs.push_back({r.rank - 1, r.rank, nullptr, rl, DdRecord<T>::bytes((size_t)from_l[q]), -1});
            if (from_r[q]) msgs.push_back({r.rank + 1, r.rank, nullptr, rr, DdRecord<T>::bytes((size_t)from_r[q]), -1});
        }
        exchange(pair_up(msgs), false);
        for (int q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            if (kill) { if (n_l[q]) e.dd_kill(idx_l[q], n_l[q]); if (n_r[q]) e.dd_kill(idx_r[q], n_r[q]); }
            if (from_l[q]) e.dd_append(r.rec_r[0].p, from_l[q], flag_from_left);
            if (from_r[q]) e.dd_append(r.rec_r[1].p, from_r[q], flag_from_right);
            // the send buffers are reused by the next migrate: everything queued so far must have read them
            HC(hipStreamSynchronize(r.main));
        }
        if (!use_rccl) for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); }
        if (got_l) *got_l = from_l;
        if (got_r) *got_r = from_r;
    }
    // Posted halves {src → dst: sbuf} and {src → dst: rbuf} become one message per (src, dst) when both ends are local;
    // with RCCL the two halves stay separate calls anyway (Send on the source's communicator, Recv on the target's).
    std::vector<Msg> pair_up(const std::vector<Msg>& posted) {
        std::vector<Msg> out;
        for (const Msg& m : posted) {
            bool merged = false;
            for (Msg& o : out) if (o.src == m.src && o.dst == m.dst && o.slot == m.slot) {
                if (m.sbuf) o.sbuf = m.sbuf;
                if (m.rbuf) o.rbuf = m.rbuf;
                merged = true; break;
            }
            if (!merged) out.push_back(m);
        }
        if (use_rccl) {
            // one entry per local END: a message whose both ends are local is sent by one and received by the other —
            // exchange() issues Send when the source is local and Recv when the target is local, which is what `out` holds
            return out;
        }
        return out;
    }

    // ---- one neighbour pass with its halo, all local ranks, phase by phase ----------------------------------------
    void halo_msgs(int which, std::vector<Msg>& msgs) {
        msgs.clear();
        const size_t vb = 2 * sizeof(V4);
        for (auto& r : R) {
            Halo& h = r.halo[which];
            if (r.has_left && h.n_send_l) msgs.push_back({r.rank, r.rank - 1, h.sb_l.p, nullptr, (size_t)h.n_send_l * vb, which * 4 + 0});
            if (r.has_right && h.n_send_r) msgs.push_back({r.rank, r.rank + 1, h.sb_r.p, nullptr, (size_t)h.n_send_r * vb, which * 4 + 1});
            if (r.has_left && h.n_slot_l) msgs.push_back({r.rank - 1, r.rank, nullptr, h.rb_l.p, (size_t)h.n_slot_l * vb, which * 4 + 1});
            if (r.has_right && h.n_slot_r) msgs.push_back({r.rank + 1, r.rank, nullptr, h.rb_r.p, (size_t)h.n_slot_r * vb, which * 4 + 0});
        }
    }
    // $SPHMI_DD_ONE_SLAB_AT_A_TIME=1 (measurement only, tools/slab_pass_time.py): the slabs of a one-process handle that share ONE GPU take their passes one after
    // the other — all slabs pack, the messages are copied, then slab by slab: interior launch on the main stream ‖ unpack + slab-edge launch on the side stream,
    // and the host waits for both before the next slab starts.  A slab's own two launches still overlap each other, but no other slab's work is on the chip:
    // the time between the two host synchronisations is what a GPU of its own would spend on the pass with the halo already landed.  `pass_us[q][which-1]` adds it up.
    bool one_at_a_time = false;
    std::vector<std::array<double, 2>> pass_us; std::vector<std::array<long long, 2>> pass_n;
    void pass_one_at_a_time(int which) {
        const int set = which - 1;
        std::vector<Msg> msgs;
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            if (moving) e.dd_progress_motion();
            Halo& h = r.halo[set];
            const int n = h.n_send_l + h.n_send_r;
            wait_consumed(r, set * 4 + 0); wait_consumed(r, set * 4 + 1);
            if (n) {
                const int s_ = set == 0 ? e.iA : e.iH;
                hipLaunchKernelGGL(k_halo_pack2<T>, dim3((n + 255) / 256), dim3(256), 0, r.main, e.pk0[s_], e.pk1[s_],
                                   (const int*)h.send_l.p, h.n_send_l, (V4*)h.sb_l.p, (const int*)h.send_r.p, h.n_send_r, (V4*)h.sb_r.p);
                HC(hipGetLastError());
            }
            HC(hipEventRecord(r.ev_pack, r.main));
        }
        halo_msgs(set, msgs);
        exchange(pair_up(msgs), false);
        for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); HC(hipStreamSynchronize(r.side)); }
        if (pass_us.size() != R.size()) { pass_us.assign(R.size(), {0.0, 0.0}); pass_n.assign(R.size(), {0, 0}); }
        for (size_t q = 0; q < R.size(); ++q) {
            Rank& r = R[q];
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            Halo& h = r.halo[set];
            const int n = h.n_slot_l + h.n_slot_r;
            const auto t0 = std::chrono::steady_clock::now();
            HC(hipEventRecord(r.ev_pack, r.main));
            HC(hipStreamWaitEvent(r.side, r.ev_pack, 0));
            e.dd_pass(which, 0.0, 1);
            e.stream = r.side;
            try {
                if (n) {
                    const int s_ = set == 0 ? e.iA : e.iH;
                    hipLaunchKernelGGL(k_halo_unpack2<T>, dim3((n + 255) / 256), dim3(256), 0, r.side, e.pk0[s_], e.pk1[s_],
                                       (const int*)h.slot_l.p, h.n_slot_l, (const V4*)h.rb_l.p, (const int*)h.slot_r.p, h.n_slot_r, (const V4*)h.rb_r.p);
                    HC(hipGetLastError());
                }
                e.dd_pass(which, 0.0, 2);
                HC(hipEventRecord(r.ev_edge, r.side));
                HC(hipStreamWaitEvent(r.main, r.ev_edge, 0));
            } catch (...) { e.stream = r.main; throw; }
            e.stream = r.main;
            HC(hipStreamSynchronize(r.main));
            pass_us[q][set] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            pass_n[q][set] += 1;
        }
    }
    void pass(int which) {
        if (one_at_a_time && cfg.mdbc == SPHMI_MDBC_NONE && !use_rccl && !shm) { pass_one_at_a_time(which); return; }
        const int set = which - 1;
        bool serial = !overlap || (cfg.mdbc == SPHMI_MDBC_SIMPLE && which == 1);
        // (no local slab sends or receives anything — a one-slab world: the whole pass is one launch on the main stream, without the
        // event hand-overs to the side stream, each of which is a bubble of ≈6 µs in the main stream)
        {
            bool any = false;
            for (auto& r : R) { const Halo& h = r.halo[set]; any |= (h.n_send_l + h.n_send_r + h.n_slot_l + h.n_slot_r) != 0; }
            if (!any && world == 1) serial = true;
        }
        std::vector<Msg> msgs;
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            if (moving) e.dd_progress_motion();                          // :765 / :787 — owned and ghost copies move alike, then the pack
            Halo& h = r.halo[set];
            const int n = h.n_send_l + h.n_send_r;
            wait_consumed(r, set * 4 + 0); wait_consumed(r, set * 4 + 1);
            if (n) {
                const int s_ = set == 0 ? e.iA : e.iH;
                hipLaunchKernelGGL(k_halo_pack2<T>, dim3((n + 255) / 256), dim3(256), 0, r.main, e.pk0[s_], e.pk1[s_],
                                   (const int*)h.send_l.p, h.n_send_l, (V4*)h.sb_l.p, (const int*)h.send_r.p, h.n_send_r, (V4*)h.sb_r.p);
                HC(hipGetLastError());
            }
            HC(hipEventRecord(r.ev_pack, r.main));
            if (!serial) {
                HC(hipStreamWaitEvent(r.side, r.ev_pack, 0));            // everything queued so far: the previous pass, the pack
                e.dd_pass(which, 0.0, 1);                                // interior tiles: need owned data only
            }
        }
        halo_msgs(set, msgs);
        exchange(pair_up(msgs), !serial);
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            Halo& h = r.halo[set];
            const int n = h.n_slot_l + h.n_slot_r;
            hipStream_t q = serial ? r.main : r.side;
            e.stream = q;
            try {
                if (n) {
                    const int s_ = set == 0 ? e.iA : e.iH;
                    hipLaunchKernelGGL(k_halo_unpack2<T>, dim3((n + 255) / 256), dim3(256), 0, q, e.pk0[s_], e.pk1[s_],
                                       (const int*)h.slot_l.p, h.n_slot_l, (const V4*)h.rb_l.p, (const int*)h.slot_r.p, h.n_slot_r, (const V4*)h.rb_r.p);
                    HC(hipGetLastError());
                }
                if (serial) {
                    // mDBC (:772) reads the fluid of state A in the ghost layers and rewrites the boundary densities that
                    // every tile of pass 1 may read: halo → mDBC → the whole pass, nothing to overlap
                    if (cfg.mdbc == SPHMI_MDBC_SIMPLE && which == 1) e.dd_mdbc();
                    e.dd_pass(which, 0.0, 0);
                } else {
                    e.dd_pass(which, 0.0, 2);                            // slab-edge tiles, as soon as the halo has landed
                    HC(hipEventRecord(r.ev_edge, r.side));
                    HC(hipStreamWaitEvent(r.main, r.ev_edge, 0));         // the next pack / the reductions need the edge tiles
                }
            } catch (...) { e.stream = r.main; throw; }
            e.stream = r.main;
        }
    }

    // local maxima → global maxima → decisions, without leaving the device
    void reductions_and_control() {
        const int p = parity; parity ^= 1;          // the set of slots the LAST corrector filled; the coming one fills the other
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            r.e->serve_reschedules();
            if (!use_rccl) HC(hipEventRecord(r.ev_red, r.main));         // behind the last corrector (its edge tiles joined `main`)
        }
        if (mbox_on) {
            // post + wait + merge + control in ONE launch per slab (k_dd_mbox_merge_control); the host only counts the steps it queued
            mbox_seq += 1;
            for (auto& r : R) {
                HC(hipSetDevice(r.device));
                MboxArgs A{};
                for (int q = 0; q < world && q < 16; ++q) A.dst[q] = r.box_of[q];
                // (the box parity follows the SEQUENCE number, not the step parity: upload() restarts the step parity while the sequence keeps counting, and two
                // consecutive posts must never share a half of the box)
                A.own = r.box; A.mine = r.e->red_d + 4 * p; A.world = world; A.me = r.rank; A.parity = (int)(mbox_seq & 1ull); A.seq = mbox_seq; A.timeout = mbox_timeout;
                hipLaunchKernelGGL(k_dd_mbox_merge_control<T>, dim3(1), dim3(64), 0, r.main, r.M, A, r.e->red_d + 4 * (p ^ 1), r.e->ctrl_d, cfg.h, cfg.c0, cfg.CFL);
                HC(hipGetLastError());
                r.e->dd_control_queued(p ^ 1);
            }
            return;
        }
        if (use_rccl && world > 1) {
            Rccl& N = Rccl::get();
            NCX("per-step allreduce", R[0].rank, -1, R[0].comm_red, N.GroupStart());
            for (auto& r : R) {
                HC(hipSetDevice(r.device));
                unsigned long long* t = r.e->red_d + 4 * p;
                NCX("per-step allreduce (4 × uint64, max)", r.rank, -1, r.comm_red, N.AllReduce(t, t, 4, ncclUint64, ncclMax, r.comm_red, r.main));
            }
            NCX("per-step allreduce", R[0].rank, -1, R[0].comm_red, N.GroupEnd());
        }
        if (shm) {
            Rank& r = R[0];
            unsigned long long* t = r.e->red_d + 4 * p;
            HC(hipMemcpyAsync(red_h, t, 32, hipMemcpyDeviceToHost, r.main));
            HC(hipStreamSynchronize(r.main));
            shm->allreduce((int64_t*)red_h, 4, ShmWorld::MAXU);
            HC(hipMemcpyAsync(t, red_h, 32, hipMemcpyHostToDevice, r.main));
            HC(hipStreamSynchronize(r.main));
        }
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            RedPtrs P{}; P.n = 0;
            P.p[P.n++] = r.e->red_d + 4 * p;
            if (!use_rccl) {
                // Double-buffered by step parity: slab q's set p is zeroed again by ITS control two steps from now, which waits for
                // every slab's corrector of the step in between — queued behind that slab's control of this step, the reader.
                for (auto& o : R) {
                    if (&o == &r) continue;
                    HC(hipStreamWaitEvent(r.main, o.ev_red, 0));
                    if (o.device == r.device) P.p[P.n++] = o.e->red_d + 4 * p;
                    else {
                        HC(hipMemcpyAsync(r.stage + 4 * o.rank, o.e->red_d + 4 * p, 32, hipMemcpyDeviceToDevice, r.main));
                        P.p[P.n++] = r.stage + 4 * o.rank;
                    }
                }
            }
            hipLaunchKernelGGL(k_dd_merge_control<T>, dim3(1), dim3(64), 0, r.main, r.M, P, r.e->red_d + 4 * (p ^ 1), r.e->ctrl_d, cfg.h, cfg.c0, cfg.CFL);
            HC(hipGetLastError());
            r.e->dd_control_queued(p ^ 1);
        }
    }

    // ---- the SimulationLoop of src/SPHCellList.jl:727-805, over all slabs --------------------------------------------
    void advance(double t_target, int64_t max_steps, sphmi_progress* out) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_advance before sphmi_upload");
        double dxl = 1.0 + cfg.h;                                        // :739
        int64_t steps = 0;
        // the first iteration of the loop always rebuilds (:758): run it now when there will be one (StepCtrl::pre_rebuilt) instead of
        // a one-step batch — allreduce, four messages — that the control cancels
        const bool pre = total_time <= t_target && max_steps != 0;
        sphmi_dd_control st{};
        try {
            if (pre) rebuild_collective();
            for (auto& r : R) { HC(hipSetDevice(r.device)); r.e->total_time = total_time; r.e->last_dt = last_dt; r.e->dd_ctrl_init(dxl, t_target, max_steps, pre); }
            if (pre) dxl = 0.0;
            bool fresh = pre;            // the first queued step follows a rebuild: it adds nothing to Δx (Engine::advance)
            for (;;) {
                // A step that asks for a rebuild cancels the rest of its batch, and a cancelled step still pays its allreduce
                // and its messages: queue up to the step EXPECTED to ask (Δx grows by 4·max|Δx| a step, slowly changing).
                int batch = kBatch;
                if (dx_rate > 0.0) batch = std::max(1, std::min(batch, (int)std::ceil((cfg.h - dxl) / dx_rate) + (fresh ? 1 : 0)));
                if (max_steps >= 0) batch = (int)std::max<int64_t>(1, std::min<int64_t>(batch, max_steps - steps));
                const double dx0 = dxl; const int64_t steps0 = steps;
                for (int k = 0; k < batch; ++k) {
                    reductions_and_control();
                    if (have_halo) { pass(1); pass(2); }                 // before the first rebuild there is no ghost layer to exchange
                }
                for (auto& r : R) {
                    HC(hipSetDevice(r.device)); sphmi_dd_control s{}; r.e->dd_ctrl_sync(&s);
                    if (&r == &R[0]) st = s;
                    else if (s.steps_done != st.steps_done || s.need_rebuild != st.need_rebuild || s.stop != st.stop || s.error != st.error ||
                             memcmp(&s.total_time, &st.total_time, sizeof(double)) != 0) {
                        // every slab takes the same decisions from the same merged maxima: a difference is a bug of the driver, never a
                        // state to hand out
                        char buf[200];
                        snprintf(buf, sizeof buf, "slab %d is out of step with slab %d (steps %lld / %lld, t %.17g / %.17g, rebuild %d / %d, stop %d / %d)",
                                 r.rank, R[0].rank, (long long)s.steps_done, (long long)st.steps_done, s.total_time, st.total_time, s.need_rebuild, st.need_rebuild, s.stop, st.stop);
                        throw EngineError(SPHMI_ERR_STATE, buf);
                    }
                }
                steps = st.steps_done;
                total_time = st.total_time; last_dt = st.last_dt; dxl = st.delta_x;
                const int64_t grown = (steps - steps0) + (st.need_rebuild ? 1 : 0) - (fresh && steps > steps0 ? 1 : 0);
                if (steps > steps0) fresh = false;
                if (grown > 0 && dx0 < cfg.h && st.delta_x > dx0) dx_rate = (st.delta_x - dx0) / (double)grown;
                if (st.error == 4) throw EngineError(SPHMI_ERR_DEVICE, "SPHMI_EXCHANGE=mailbox: a peer slab did not post its maxima in time ($SPHMI_MBOX_TIMEOUT seconds)");
                if (st.error == 2) throw EngineError(SPHMI_ERR_NUMERIC, "non-positive density produced on some slab");
                if (st.error) throw EngineError(SPHMI_ERR_NUMERIC, "non-positive or NaN dt");
                if (st.need_rebuild) {
                    rebuild_collective();
                    dxl = 0.0;
                    for (auto& r : R) { HC(hipSetDevice(r.device)); r.e->dd_ctrl_resume(); }
                    fresh = true;
                    continue;
                }
                if (st.stop || !(total_time <= t_target) || (max_steps >= 0 && steps >= max_steps)) break;
            }
        } catch (...) { iteration += steps; delta_x = dxl; fill(out, steps); throw; }
        iteration += steps; delta_x = dxl;
        for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); }
        {   // the bad-ρ flag of the LAST corrector: no control looks at it in this call (Engine::advance does the same)
            std::vector<long long> bad(R.size(), 0);
            for (size_t q = 0; q < R.size(); ++q) {
                Rank& r = R[q]; HC(hipSetDevice(r.device));
                unsigned long long f = 0;
                r.e->bounce.d2h(&f, r.e->red_cur() + 3, 8, r.main);
                bad[q] = f != 0;
            }
            host_allreduce(bad, 1, OP_MAX);
            if (bad[0]) { fill(out, steps); throw EngineError(SPHMI_ERR_NUMERIC, "non-positive density produced on some slab"); }
        }
        index_counter = 1;                                               // SimMetaData.IndexCounter = 1 + occupied cells (:145-157)
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            if (!e.have_grid) continue;
            HC(hipMemsetAsync(r.mm_d, 0, 4, r.main));
            hipLaunchKernelGGL(k_dd_count_owned_cells, dim3(std::min((e.N + 255) / 256, 512)), dim3(256), 0, r.main, (const int*)e.key[e.cur], (const uint8_t*)e.type[e.cur], e.N, e.grid.ncell, r.mm_d);
            HC(hipMemcpyAsync(r.mm_h, r.mm_d, 4, hipMemcpyDeviceToHost, r.main));
            HC(hipStreamSynchronize(r.main));
            index_counter += r.mm_h[0];
        }
        if (rank_mode) { std::vector<long long> v{index_counter - 1}; host_allreduce(v, 1, OP_SUM); index_counter = v[0] + 1; }
        fill(out, steps);
    }
    void fill(sphmi_progress* out, int64_t steps) {
        if (!out) return;
        out->iteration = iteration; out->steps_done = steps; out->n_rebuilds = n_rebuilds; out->index_counter = index_counter;
        out->total_time = total_time; out->last_dt = last_dt; out->delta_x = delta_x;
    }

    // ---- output side: owned particles of every local slab, merged into the order of the UNSPLIT sort ------------------
    int64_t owned_count_impl() {
        int64_t n = 0;
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            HC(hipStreamSynchronize(r.main));
            std::vector<uint8_t> ty((size_t)e.N);
            e.bounce.d2h(ty.data(), e.type[e.cur], (size_t)e.N, r.main);
            for (auto t : ty) n += (t != 0 && !(t & kGhostMask)) ? 1 : 0;
        }
        return n;
    }
    // Merged row o = row i of local slab q, for the rows the slabs OWN, in the order ONE engine would hold: a k-way merge by
    // order tag = (cell z, y, x, rank in cell) of the last sort.  ty / tag: the slabs' type bytes and tags (n[q] entries).
    template <class F>
    size_t merged_rows(const std::vector<const uint8_t*>& ty, const std::vector<const unsigned long long*>& tag,
                       const std::vector<size_t>& n, F&& put) {
        const size_t L = R.size();
        std::vector<size_t> at(L, 0);
        auto skip = [&](size_t q) { while (at[q] < n[q] && (ty[q][at[q]] == 0 || (ty[q][at[q]] & kGhostMask))) ++at[q]; };
        for (size_t q = 0; q < L; ++q) skip(q);
        size_t o = 0;
        const size_t cap = (size_t)cfg.n_particles;
        for (;;) {
            int best = -1;
            for (size_t q = 0; q < L; ++q) if (at[q] < n[q] && (best < 0 || tag[q][at[q]] < tag[(size_t)best][at[(size_t)best]])) best = (int)q;
            if (best < 0) break;
            if (o >= cap) throw EngineError(SPHMI_ERR_STATE, "sphmi_download: more owned particles than the handle was created for");
            put(o, (size_t)best, at[(size_t)best]);
            ++o; ++at[(size_t)best]; skip((size_t)best);
        }
        return o;
    }
    // Asynchronous like the one-device handle's: `begin` snapshots every slab on ITS device in stream order
    // and starts the device → host copies on the slabs' copy streams, into page-locked staging the handle owns; the caller
    // may advance at once; `end` waits for the copies and merges the slabs into the caller's arrays.  Registering the
    // caller's arrays (sphmi_host_register) gains nothing here — they only ever see host copies — and is accepted as a no-op.
    struct Stage { HostBuf pos, vel, acc, rho, prs, gho, id, cel, grp, ty, tag; size_t n = 0; };
    std::vector<Stage> stage;
    struct PendingDownload { void *pos, *vel, *acc, *rho, *prs, *gho; int64_t *ids, *cells; uint8_t* ty; uint64_t* grp; bool on = false; } pend{};
    void download_begin(void* position, void* velocity, void* acceleration, void* density, void* pressure, int64_t* ids,
                        uint8_t* ty, uint64_t* groups, void* ghost_points, int64_t* cells) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_download before sphmi_upload");
        if (pend.on) download_end();
        const size_t hb = (size_t)cfg.host_float_bytes, C = (size_t)out_comp;
        stage.resize(R.size());
        for (size_t q = 0; q < R.size(); ++q) {
            Rank& r = R[q]; Stage& P = stage[q];
            HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            const size_t n = (size_t)e.N; P.n = n;
            e.set_output_components(out_comp);
            auto buf = [&](HostBuf& b, bool wanted, size_t bytes) -> void* { return wanted ? (void*)b.need(std::max<size_t>(bytes, 1)) : nullptr; };
            e.dl_tags_host = (unsigned long long*)P.tag.need(std::max<size_t>(n * 8, 1));
            e.dl_dst_page_locked = true;               // (every destination below is a HostBuf: hipHostMalloc)
            try {
                e.download_begin(buf(P.pos, position, n * C * hb), buf(P.vel, velocity, n * C * hb), buf(P.acc, acceleration, n * C * hb),
                                 buf(P.rho, density, n * hb), buf(P.prs, pressure, n * hb), (int64_t*)buf(P.id, ids, n * 8),
                                 (uint8_t*)buf(P.ty, true, n), (uint64_t*)buf(P.grp, groups, n * 8), buf(P.gho, ghost_points, n * C * hb),
                                 (int64_t*)buf(P.cel, cells, n * (size_t)D * 8));
            } catch (...) { e.dl_tags_host = nullptr; e.dl_dst_page_locked = false; throw; }
            e.dl_tags_host = nullptr; e.dl_dst_page_locked = false;
        }
        pend = PendingDownload{position, velocity, acceleration, density, pressure, ghost_points, ids, cells, ty, groups, true};
    }
    void download_end() override {
        if (!pend.on) return;
        pend.on = false;
        for (auto& r : R) { HC(hipSetDevice(r.device)); r.e->download_end(); }
        const size_t hb = (size_t)cfg.host_float_bytes, C = (size_t)out_comp, L = R.size();
        std::vector<const uint8_t*> tys(L); std::vector<const unsigned long long*> tags(L); std::vector<size_t> ns(L);
        for (size_t q = 0; q < L; ++q) { tys[q] = (const uint8_t*)stage[q].ty.p; tags[q] = (const unsigned long long*)stage[q].tag.p; ns[q] = stage[q].n; }
        const PendingDownload d = pend;
        n_downloaded = (int64_t)merged_rows(tys, tags, ns, [&](size_t o, size_t q, size_t i) {
            Stage& P = stage[q];
            auto put = [&](void* dst, HostBuf& src, size_t elem) { if (dst) memcpy((char*)dst + o * elem, src.p + i * elem, elem); };
            put(d.pos, P.pos, C * hb); put(d.vel, P.vel, C * hb); put(d.acc, P.acc, C * hb);
            put(d.rho, P.rho, hb); put(d.prs, P.prs, hb); put(d.gho, P.gho, C * hb);
            put(d.ids, P.id, 8); put(d.grp, P.grp, 8); put(d.cells, P.cel, (size_t)D * 8);
            if (d.ty) d.ty[o] = (uint8_t)(((const uint8_t*)P.ty.p)[i] & kTypeMask);
        });
    }
    void download(void* position, void* velocity, void* acceleration, void* density, void* pressure, int64_t* ids,
                  uint8_t* ty, uint64_t* groups, void* ghost_points, int64_t* cells) override {
        download_begin(position, velocity, acceleration, density, pressure, ids, ty, groups, ghost_points, cells);
        download_end();
    }
    int64_t n_downloaded = 0;
    int out_comp = 0;
    void set_output_components(int c) override {
        if (c != D && c != 3) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_set_output_components: dims or 3");
        out_comp = c;
    }
    void host_register(void* p, size_t bytes) override { if (!p && bytes) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_host_register: null pointer"); }
    void host_unregister(void*) override {}
    // one per-particle device array of every local slab (elem bytes per row), merged like a download → out
    template <class Src>
    void merged_array(Src&& src, size_t elem, const std::function<void(size_t, const char*)>& put) {
        const size_t L = R.size();
        std::vector<std::vector<char>> data(L); std::vector<std::vector<uint8_t>> ty(L); std::vector<std::vector<unsigned long long>> tag(L);
        std::vector<const uint8_t*> tys(L); std::vector<const unsigned long long*> tags(L); std::vector<size_t> ns(L);
        for (size_t q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            HC(hipStreamSynchronize(r.main));
            const size_t n = (size_t)e.N; ns[q] = n;
            data[q].resize(std::max<size_t>(n * elem, 1)); ty[q].resize(std::max<size_t>(n, 1)); tag[q].resize(std::max<size_t>(n, 1));
            if (n) {
                e.bounce.d2h(data[q].data(), src(e), n * elem, r.main);
                e.bounce.d2h(ty[q].data(), e.type[e.cur], n, r.main);
                e.bounce.d2h(tag[q].data(), e.otag[e.cur], n * 8, r.main);
            }
            tys[q] = ty[q].data(); tags[q] = tag[q].data();
        }
        n_downloaded = (int64_t)merged_rows(tys, tags, ns, [&](size_t o, size_t q, size_t i) { put(o, data[q].data() + i * elem); });
    }
    void put_packets(void* vec_out, void* scalar_out) {      // helper of the two hooks below: rows of V4 { vector, scalar }
        const bool h8 = cfg.host_float_bytes == 8;
        const int Dd = D;
        return merged_array([&](Engine<T>& e) -> const void* { return packet_src(e); }, sizeof(V4), [=](size_t o, const char* p) {
            V4 v; memcpy(&v, p, sizeof v);
            const T c[3] = {v.x, v.y, v.z};
            if (vec_out) for (int d = 0; d < Dd; ++d) { if (h8) ((double*)vec_out)[o * Dd + d] = (double)c[d]; else ((float*)vec_out)[o * Dd + d] = (float)c[d]; }
            if (scalar_out) { if (h8) ((double*)scalar_out)[o] = (double)v.w; else ((float*)scalar_out)[o] = (float)v.w; }
        });
    }
    std::function<const void*(Engine<T>&)> packet_src;
    // Parity hook (sphmi_forces_once) on slabs: the collective rebuild sorts every slab, migrates and refreshes the ghost
    // layers from the owners' current state; Pressure! → [mDBC on everything held] → one forces-only pass over the interior
    // AND the slab-edge tiles; rows come back in the order one engine would hold (rank mode: this process's slab).
    void forces_once(int apply_mdbc, void* drhodt, void* acceleration) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_forces_once before sphmi_upload");
        for (auto& r : R) { HC(hipSetDevice(r.device)); HC(hipStreamSynchronize(r.main)); }
        rebuild_collective();
        for (auto& r : R) { HC(hipSetDevice(r.device)); r.e->forces_local(apply_mdbc, true); }
        for (auto& r : R) { HC(hipSetDevice(r.device)); r.e->sync_and_collect(); }
        packet_src = [](Engine<T>& e) -> const void* { return e.rec[e.iB]; };
        put_packets(acceleration, drhodt);
    }
    // StoreKernelOutput (src/SPHCellList.jl:106-116): Σ∇W, ΣW of the last corrector pass, merged like a download
    void download_kernel_output(void* kernel, void* kernel_gradient) override {
        if (cfg.kernel_output != SPHMI_KOUT_STORE) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_kernel_output: the handle was not created with kernel_output = STORE");
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_kernel_output before sphmi_upload");
        packet_src = [](Engine<T>& e) -> const void* { return e.kout_d; };
        put_packets(kernel_gradient, kernel);
    }
    // sphmi_download_permutation: every particle carries its row at the previous call (the upload order at first) as a column of
    // its own — Engine::prow, permuted by the sorts and carried by the migration and ghost-layer records — exactly like the
    // one-device handle.  (Round 3 matched the ID column of now against the ID column of then, which is wrong without a word when
    // the caller's IDs repeat: the reference reads Idp per CSV file and concatenates, src/PreProcess.jl:28,71.)  Merged by order
    // tag like a download; the slabs' columns are then re-numbered with the merged rows.  An asynchronous download in flight is
    // not disturbed.
    void download_permutation(int64_t* prev_row) override {
        if (!uploaded) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_permutation before sphmi_upload");
        if (!prev_row) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_download_permutation: null array");
        if (rank_mode) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_permutation: one-process handles only (a rank-mode process holds one slab of the rows)");
        if (cfg.n_particles > (int64_t)INT32_MAX) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_download_permutation: more than 2^31 - 1 rows");
        const size_t N = (size_t)cfg.n_particles, L = R.size();
        std::vector<std::vector<int>> rows(L); std::vector<std::vector<uint8_t>> ty(L); std::vector<std::vector<unsigned long long>> tag(L);
        std::vector<const uint8_t*> tys(L); std::vector<const unsigned long long*> tags(L); std::vector<size_t> ns(L);
        for (size_t q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            Engine<T>& e = *r.e;
            HC(hipStreamSynchronize(r.main));
            const size_t n = (size_t)e.N; ns[q] = n;
            rows[q].resize(std::max<size_t>(n, 1)); ty[q].resize(std::max<size_t>(n, 1)); tag[q].resize(std::max<size_t>(n, 1));
            if (n) {
                e.bounce.d2h(rows[q].data(), e.prow[e.cur], n * 4, r.main);
                e.bounce.d2h(ty[q].data(), e.type[e.cur], n, r.main);
                e.bounce.d2h(tag[q].data(), e.otag[e.cur], n * 8, r.main);
            }
            tys[q] = ty[q].data(); tags[q] = tag[q].data();
        }
        std::vector<uint8_t> seen(N, 0);
        const size_t got = merged_rows(tys, tags, ns, [&](size_t o, size_t q, size_t i) {
            const int64_t v = rows[q][i];
            if (v < 0 || (size_t)v >= N || seen[(size_t)v]) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_permutation: the row column of the slabs is not a permutation");
            seen[(size_t)v] = 1;
            prev_row[o] = v;
            rows[q][i] = (int)o;                                        // the row NOW: what the next call hands out
        });
        if (got != N) throw EngineError(SPHMI_ERR_STATE, "sphmi_download_permutation: the handle does not hold the uploaded particle set");
        for (size_t q = 0; q < L; ++q) {
            Rank& r = R[q]; HC(hipSetDevice(r.device));
            if (ns[q]) r.e->bounce.h2d(r.e->prow[r.e->cur], rows[q].data(), ns[q] * 4, r.main);      // (ghost copies keep a stale row: they die at the next rebuild)
        }
    }
    void unique_cells(int64_t* out, int64_t cap, int64_t* n_out) override {
        // occupied cells of the owned particles in sort order, from the merged Cells column
        const size_t N = (size_t)cfg.n_particles;
        std::vector<int64_t> cel(N * (size_t)D);
        download(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cel.data());
        int64_t n = 0;
        for (int64_t i = 0; i < n_downloaded; ++i) {
            bool fresh = i == 0;
            for (int d = 0; d < D && !fresh; ++d) fresh = cel[(size_t)i * D + d] != cel[(size_t)(i - 1) * D + d];
            if (!fresh) continue;
            if (out) { if (n >= cap) throw EngineError(SPHMI_ERR_ARGUMENT, "sphmi_unique_cells: capacity too small"); for (int d = 0; d < D; ++d) out[n * D + d] = cel[(size_t)i * D + d]; }
            ++n;
        }
        if (n_out) *n_out = n;
    }
    // the slabs run side by side: a phase costs what its slowest slab spent in it (calls: slab 0's — the same on every slab)
    void timers(int32_t cap, const char** names, double* secs, int64_t* calls, int32_t* n) override {
        R[0].e->timers(cap, names, secs, calls, n);
        if (!secs) return;
        for (size_t q = 1; q < R.size(); ++q) {
            double sq[PH_COUNT + 1] = {}; int32_t nq = 0;
            R[q].e->timers(PH_COUNT, nullptr, sq, nullptr, &nq);
            for (int i = 0; i < PH_COUNT && i < cap; ++i) secs[i] = std::max(secs[i], sq[i]);
        }
    }
    void force_stats(int reset, double* avg_ms, int64_t* launches) override {
        double ms = 0; int64_t ln = 0;
        for (auto& r : R) { double a = 0; int64_t l = 0; r.e->force_stats(reset, &a, &l); ms += a * (double)l; ln += l; }
        if (avg_ms) *avg_ms = ln ? ms / (double)ln : 0.0;
        if (launches) *launches = ln;
    }
    void device_ptrs(void** p0, void** p1, int64_t* n) override { R[0].e->device_ptrs(p0, p1, n); }      // (slab 0 of this handle)
    // test hook (sphmi_multi_column_cost): the work histogram the re-cut balances, summed over the local slabs
    void column_cost(int64_t col0, int32_t ncols, uint64_t* out) {
        for (int c = 0; c < ncols; ++c) out[c] = 0;
        std::vector<unsigned long long> h((size_t)ncols);
        for (auto& r : R) {
            HC(hipSetDevice(r.device));
            unsigned long long* cd = (unsigned long long*)r.cost.need((size_t)ncols * 8);
            HC(hipMemsetAsync(cd, 0, (size_t)ncols * 8, r.main));
            r.e->dd_column_cost(col0, ncols, (uint64_t*)cd);
            r.e->bounce.d2h(h.data(), cd, (size_t)ncols * 8, r.main);
            for (int c = 0; c < ncols; ++c) out[c] += h[(size_t)c];
        }
    }
    // test / measurement hook (sphmi_multi_halo_info): what the scaling prediction of DESIGN §7 is computed from — per local slab
    // { slab, rows held (owned + ghost copies), halo records sent left / right with state A, with the half-step state H, tiles of the interior launch, tiles of the
    //   slab-edge launch, blocks per XCD run of the two launches }
    static constexpr int kHaloInfoWords = 12;    // (words 10, 11: $SPHMI_DD_ONE_SLAB_AT_A_TIME — mean host-timed nanoseconds of the slab's pass 1 / pass 2 run alone on the chip, 0 otherwise)
    int halo_info(int64_t* out, int cap_words) {
        int k = 0;
        for (auto& r : R) {
            if (!r.e || k + kHaloInfoWords > cap_words) break;
            const size_t q = (size_t)(&r - &R[0]);
            const bool timed = one_at_a_time && q < pass_us.size();
            const int64_t v[kHaloInfoWords] = {r.rank, r.e->N, r.halo[0].n_send_l, r.halo[0].n_send_r, r.halo[1].n_send_l, r.halo[1].n_send_r,
                                               r.e->list_tiles[0], r.e->list_tiles[1], r.e->part_max[0], r.e->part_max[1],
                                               timed && pass_n[q][0] ? (int64_t)(1e3 * pass_us[q][0] / (double)pass_n[q][0]) : 0,
                                               timed && pass_n[q][1] ? (int64_t)(1e3 * pass_us[q][1] / (double)pass_n[q][1]) : 0};
            for (int i = 0; i < kHaloInfoWords; ++i) out[k + i] = v[i];
            k += kHaloInfoWords;
        }
        return k;
    }
    void reset_count() override {}
    int64_t owned_count() override { return owned_count_impl(); }
    void multi_info(sphmi_multi_info* o) {
        memset(o, 0, sizeof *o);
        o->world = world; o->n_local = (int32_t)R.size(); o->axis = axis; o->halo_width = halo_width; o->n_recuts = n_recuts;
        o->transport = shm ? 2 : (use_rccl ? 1 : 0);
        o->reserved = mbox_on ? 1 : 0;                                   // how the per-step maxima travel: 0 the transport's collective, 1 mailboxes
        for (int r = 1; r < world && r < 16; ++r) o->cuts[r - 1] = plan.world() == world ? plan.lo[r] : 0;
        for (auto& r : R) if (r.e && r.rank < 16) o->n_live[r.rank] = r.e->N;
    }
};

}  // namespace sphmi
