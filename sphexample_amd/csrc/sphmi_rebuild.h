// sphmi_rebuild.h — cell-list build (UpdateNeighbors!, src/SPHCellList.jl:138-163), mDBC
// (src/SPHCellList.jl:219-266, 319-365, 598-622) and the start-up reductions, as gfx950 kernels.
//
// The reference sorts a 17-field StructArray with a stable comparison sort and keeps a
// Dict{CartesianIndex,Int}.  Here: particle → padded linear cell id (x fastest, matching
// CartesianIndex order "last axis most significant"), histogram with atomics, exclusive scan to
// cell_start over the dense bounding grid (replaces ParticleRanges + CellDict), unordered scatter,
// then an in-cell rank fix that restores the STABLE order (rank = number of cell mates with a
// smaller previous index), and one gather pass that permutes every state array.
#pragma once
#include "sphmi_kernels.h"

namespace sphmi {

// map_floor, src/SPHCellList.jl:56-61: round half away from zero, trunc(|x|·H⁻¹ + 0.5) with the
// multiply-add fused (Julia's muladd lowers to an fma on every FMA-capable CPU).
template <class T> __device__ __forceinline__ int map_floor(T x, T inv_cutoff) {
    T t;
    if constexpr (sizeof(T) == 8) t = trunc(fma(absT(x), inv_cutoff, T(0.5)));
    else t = truncf(fmaf(absT(x), inv_cutoff, T(0.5)));
    int c = (int)t;
    return x < T(0) ? -c : c;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// bbox[0..2] = min cell, bbox[3..5] = max cell (ExtractCells!, :118-123, reduced)
// Type byte: low bits = ParticleType (1/2/3), 0 = dead (left the rank at a domain-decomposition rebuild),
// 0x80 / 0x40 = ghost copy of a particle owned by the left / right neighbour rank.
constexpr uint8_t kGhostLeft = 0x80, kGhostRight = 0x40, kGhostMask = 0xC0, kTypeMask = 0x3F;

struct GridDesc {
    int gmin[3];      // cell coordinate of padded index 1
    int np[3];        // padded dims (n + 2)
    int ncell;        // np[0]*np[1]*np[2]
};

// bbox[0..5]: min / max cell coordinate per axis.  With `old_key` (the keys of the last sort, on the grid `og`): bbox[6] is set when
// some particle is no longer in the cell it was sorted into — if none is, the stable sort the reference runs here
// (src/SPHCellList.jl:142) is the identity permutation and the rebuild has nothing to do (Engine::rebuild).
template <class T, int D>
__global__ void __launch_bounds__(256) k_cell_bbox(Half<const typename Vec4<T>::type> pk0, const uint8_t* type, int N,
                                                   T inv_cutoff, int* bbox, const int* old_key, GridDesc og) {
    // grid-stride: a few hundred blocks, so that the six atomics per WAVE are a few thousand in total — one wave per
    // 64 particles (16 k waves at 1 M) spent 228 µs queueing on the one cache line, for 16 MB of streaming reads
    int mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    bool moved = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        if (type[i] == 0) continue;
        const auto p = pk0[i];
        const int c[3] = {map_floor<T>(p.x, inv_cutoff), map_floor<T>(p.y, inv_cutoff),
                          D == 3 ? map_floor<T>(p.z, inv_cutoff) : 0};
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = c[d] < mn[d] ? c[d] : mn[d]; mx[d] = c[d] > mx[d] ? c[d] : mx[d]; }
        if (old_key) {
            const int cx = c[0] - og.gmin[0] + 1, cy = c[1] - og.gmin[1] + 1, cz = D == 3 ? c[2] - og.gmin[2] + 1 : 0;
            const bool inside = cx >= 1 && cx <= og.np[0] - 2 && cy >= 1 && cy <= og.np[1] - 2 && (D < 3 || (cz >= 1 && cz <= og.np[2] - 2));
            moved |= !inside || (cx + og.np[0] * (cy + og.np[1] * cz)) != old_key[i];
        }
    }
    if (old_key && __builtin_amdgcn_ballot_w64(moved) != 0 && (threadIdx.x & 63) == 0) atomicOr(&bbox[6], 1);
    __shared__ int s_box[4][6];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int lo = wave_min_i(mn[d]), hi = wave_max_i(mx[d]);
        if ((threadIdx.x & 63) == 0) { s_box[threadIdx.x >> 6][d] = lo; s_box[threadIdx.x >> 6][3 + d] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        int v = s_box[0][d];
        for (int w = 1; w < 4; ++w) v = d < 3 ? (s_box[w][d] < v ? s_box[w][d] : v) : (s_box[w][d] > v ? s_box[w][d] : v);
        if (d < 3) atomicMin(&bbox[d], v); else atomicMax(&bbox[d], v);
    }
}

// `flags` (device-side rebuilds, Engine::rebuild_device): the grid is the one the LAST host-side rebuild chose — the bounding box of
// then plus kStickySlack cell layers — so a particle may have left it: flags[kFlagOverflow] is raised and the cell is clamped
// (the launches that follow stay memory-safe; their results are discarded: k_permute copies instead of permuting, the step control
// stops with error 3 and the host rebuilds on a new grid).
constexpr int kStickySlack = 2, kFlagOverflow = 7, kFlagOverflowSeen = 6;
// What the host would upload into the control block before the steps behind this rebuild (a 112-byte copy costs more than this
// launch's share of a small rebuild): mode 1 — the whole block (a new sphmi_advance call), mode 2 — the request has been served
// (Δx := 0, need_rebuild := 0; `resume` stays: the queued step re-uses its Δt).  Written by the first thread of the rebuild's first launch.
struct CtrlPatch { StepCtrl* dst; int mode; StepCtrl value; int* zero; };      // zero: a counter of the rebuild cleared for the launches behind (k_permute's list of ghost-bearing rows)
template <class T, int D>
__global__ void __launch_bounds__(256) k_cell_count(Half<const typename Vec4<T>::type> pk0, const uint8_t* type, int N,
                                                    T inv_cutoff, GridDesc g, int* count, int* key, int* slot, int* flags, const CtrlPatch cp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (i == 0) {
        if (cp.zero) *cp.zero = 0;
        if (cp.mode == 1) *cp.dst = cp.value;
        else if (cp.mode == 2) { cp.dst->delta_x = 0.0; cp.dst->need_rebuild = 0; }
    }
    int k = -1;                               // lanes past the end: a key no particle has
    if (i < N) {
        auto p = pk0[i];
        int cx = map_floor<T>(p.x, inv_cutoff) - g.gmin[0] + 1;
        int cy = map_floor<T>(p.y, inv_cutoff) - g.gmin[1] + 1;
        int cz = D == 3 ? map_floor<T>(p.z, inv_cutoff) - g.gmin[2] + 1 : 0;
        if (flags) {
            const bool out = cx < 1 || cx > g.np[0] - 2 || cy < 1 || cy > g.np[1] - 2 || (D == 3 && (cz < 1 || cz > g.np[2] - 2));
            if (out && type[i] != 0) {
                atomicOr(&flags[kFlagOverflow], 1);
                cx = min(max(cx, 1), g.np[0] - 2); cy = min(max(cy, 1), g.np[1] - 2);
                if (D == 3) cz = min(max(cz, 1), g.np[2] - 2);
            }
        }
        k = cx + g.np[0] * (cy + g.np[1] * cz);
        if (type[i] == 0) k = g.ncell;        // dead particles sort behind every cell ("graveyard" key)
        key[i] = k;
    }
    // The particles arrive in the previous sorted order, so a wave holds a few RUNS of equal keys: one atomic per run
    // (its first lane adds the run length and hands out base + offset) instead of 64 colliding returning atomics.
    // Any numbering of a cell's particles will do here — k_rankfix restores the stable order.
    const int kprev = __shfl_up(k, 1, 64);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(lane == 0 || k != kprev);
    const unsigned long long upto = heads & (~0ull >> (63 - lane));                 // heads at or below this lane
    const int start = 63 - __builtin_clzll(upto);                                   // first lane of my run
    const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
    const int next = above ? __builtin_ctzll(above) : 64;                           // first lane of the next run
    int base = 0;
    if (lane == start && i < N) {
        // my run ends at the next head — seen from the run's first lane
        base = atomicAdd(&count[k], next - start);
    }
    base = __shfl(base, start, 64);
    if (i < N) slot[i] = base + (lane - start);
}

// ---- exclusive scan over the cell histogram (3 launches) ------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanPer = 16;
constexpr int kScanTile = kScanThreads * kScanPer;

__global__ void __launch_bounds__(kScanThreads) k_scan_tile(const int* in, int* out, int n, int* tsum,
                                                            int* nonempty) {
    __shared__ int s_w[kScanThreads / 64];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    int v[kScanPer];
    int sum = 0, nz = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        int idx = base + k;
        int x = idx < n ? in[idx] : 0;
        v[k] = sum;
        sum += x;
        nz += x != 0;
    }
    // wave inclusive scan of the per-thread sums
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(inc, o, 64);
        if ((threadIdx.x & 63) >= o) inc += u;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nz += __shfl_xor(nz, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_w[w] = inc;
    if ((threadIdx.x & 63) == 0 && nz) atomicAdd(nonempty, nz);
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += s_w[k];
    const int excl = woff + inc - sum;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        int idx = base + k;
        if (idx < n) out[idx] = v[k] + excl;
    }
    if (threadIdx.x == kScanThreads - 1) tsum[blockIdx.x] = woff + inc;
}

__global__ void __launch_bounds__(1024) k_scan_tsums(int* tsum, int ntiles, int* total_out) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < ntiles; b0 += 1024) {
        int idx = b0 + threadIdx.x;
        int x = idx < ntiles ? tsum[idx] : 0;
        int inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int u = __shfl_up(inc, o, 64);
            if ((threadIdx.x & 63) >= o) inc += u;
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) s_w[w] = inc;
        __syncthreads();
        int woff = s_carry;
        for (int k = 0; k < w; ++k) woff += s_w[k];
        if (idx < ntiles) tsum[idx] = woff + inc - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = s_carry;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_add(int* out, int n, const int* tsum, const int* total) {
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    const int off = tsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        int idx = base + k;
        if (idx < n) out[idx] += off;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// (the same without the total written behind the array: for scans over arrays that have no slot n)
__global__ void __launch_bounds__(kScanThreads) k_scan_add_nototal(int* out, int n, const int* tsum) {
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    const int off = tsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        int idx = base + k;
        if (idx < n) out[idx] += off;
    }
}

// The same scan in ONE launch of one workgroup, for the cell histograms of the small handles (≤ kScanSingleMax entries: every stock
// example of the reference): three launches → one.  Consumes `in` — it is left zeroed for the next rebuild's k_cell_count — and
// leaves the number of non-empty entries in misc[0] and the total in misc[1] and out[n].
constexpr int kScanSingleMax = 1 << 16;
__global__ void __launch_bounds__(1024) k_scan_single(int* in, int* out, int n, int* misc) {
    __shared__ int s_w[16], s_nz[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int carry = 0, nonempty = 0;
    for (int b0 = 0; b0 < n; b0 += 4096) {
        const int base = b0 + (int)threadIdx.x * 4;
        int v[4], sum = 0, nz = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = base + k; const int x = idx < n ? in[idx] : 0; if (idx < n) in[idx] = 0; v[k] = sum; sum += x; nz += x != 0; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nz += __shfl_xor(nz, o, 64);
        if (lane == 63) s_w[w] = inc;
        if (lane == 0) s_nz[w] = nz;
        __syncthreads();
        int woff = 0, tot = 0, tnz = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int t = s_w[k]; if (k < w) woff += t; tot += t; tnz += s_nz[k]; }
        const int excl = carry + woff + inc - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = base + k; if (idx < n) out[idx] = v[k] + excl; }
        carry += tot; nonempty += tnz;
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[n] = carry; misc[0] = nonempty; misc[1] = carry; }
}

// ------------------------------------------------------------------------------------------
// Tile schedule of the neighbour kernel.  The dispatcher hands block b to XCD b % 8 and workgroups start
// in block order, so the engine decides (at every rebuild) WHICH tile each block processes:
//   * every XCD gets one contiguous run of tiles (neighbouring tiles share source rows → one L2), with
//     the run boundaries placed so that the eight runs carry equal estimated cost (the water is not
//     spread evenly over the sorted order: equal-count runs were 8 % out of balance on the dam break);
//   * inside a run the expensive tiles start first and the cheap wall tiles fill the tail.
// cost(tile) = candidates its phase 1 scans = Σ over cell rows of the union range length.
// ------------------------------------------------------------------------------------------
// Tile list a tile belongs to (domain decomposition; without a slab every tile is list 0):
//   0 interior — needs owned data only, can run while the halo is still in flight
//   1 edge     — holds an owned particle of a slab-edge cell column (its neighbourhood reaches the ghost layer)
//   2 none     — ghost copies only: nothing to compute
__global__ void __launch_bounds__(256) k_tile_class(const int* key, const uint8_t* type, int N, int ntile, int nxp,
                                                    int nyp, int axis, int col_lo_pad, int col_hi_pad, uint8_t* cls) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (t >= ntile) return;
    const int i = t * 64 + lane;
    bool owned = false, edge = false;
    if (i < N) {
        const int k = key[i];
        owned = type[i] != 0 && (type[i] & kGhostMask) == 0;
        const int col = axis == 0 ? k % nxp : (axis == 1 ? (k / nxp) % nyp : k / (nxp * nyp));
        edge = owned && (col == col_lo_pad || col == col_hi_pad);
    }
    const unsigned long long bo = __ballot(owned), be = __ballot(edge);
    if (lane == 0) cls[t] = bo == 0 ? 2 : (be != 0 ? 1 : 0);
}

// cost of a tile in its own list, 0 in the other
__global__ void __launch_bounds__(256) k_tile_cost(const int* key, const int* cstart, const uint8_t* cls, int N, int ntile,
                                                   int nxp, int nxyp, int D, int ncell, int* cost0, int* cost1) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntile) return;
    const int kf = key[t * 64], kl0 = key[min(t * 64 + 63, N - 1)];
    const int nseg = D == 3 ? 9 : 3;
    int c = 64;
    // dead particles (first sort of a slab rebuild) carry the graveyard key ncell: clamp to the last real cell, and a
    // tile that starts in the graveyard costs nothing (cstart has ncell + 2 entries)
    const int kl = min(kl0, ncell - 1);
    for (int seg = 0; seg < nseg && kf < ncell; ++seg) {
        const int off = D == 3 ? ((seg % 3) - 1) * nxp + ((seg / 3) - 1) * nxyp : (seg - 1) * nxp;
        c += cstart[min(kl + off + 2, ncell)] - cstart[max(kf + off - 1, 0)];
    }
    if (kf >= ncell) c = 0;
    const int k = cls ? cls[t] : 0;
    cost0[t] = k == 0 ? c : 0;
    cost1[t] = k == 1 ? c : 0;
}

// Cost classes of a run (most expensive first).  Four for launches of several rounds of the wave slots: inside a class the
// tiles keep their sorted order and with it the L2 locality of the run (16 / 64 classes at 1.06 M particles: −9 / −10 %).
// Sixteen for the launches that fit the chip at once (two waves per tile, below 10 k tiles): there every tile starts at
// t = 0, the dispatcher deals the blocks round the compute units in launch order, and the finer the deal follows the
// cost the more even the units' sums — 158 791 particles 7.23e8 → 7.53e8 updates/s, 517 818 particles 9.40 → 9.57e8.
#ifndef SPHMI_TILE_CLASSES
#define SPHMI_TILE_CLASSES 4
#endif
#ifndef SPHMI_TILE_CLASSES_ONE_ROUND
#define SPHMI_TILE_CLASSES_ONE_ROUND 16
#endif
// one workgroup per XCD run: find the run, then a STABLE partition of its tiles into cost classes, most
// expensive class first.  Inside a class the tiles keep their sorted order, so the ~1000 tiles an XCD has
// in flight at any time are still neighbours in space and share their source rows in its L2 (a full sort
// by cost tripled the HBM fetch of the neighbour kernel).
// cost: this list's tile costs (0 = tile not in the list); cscan: their exclusive scan, ntile + 1 entries
struct XcdShares { float cum[9]; };      // cumulative share of the estimated cost per XCD: cum[0] = 0 … cum[8] = 1
// `tail_permille` > 0 (round 6): the LAST tail_permille / 1000 of a run are ordered again, by `tail_classes` cost classes of THEIR cost range, most expensive first.
// A launch ends when its last wave does: with four coarse classes the last tiles an XCD starts still include tiles of 40–70 µs, and the run's slots empty while
// they finish (tools/trace_tiles.py: the last 10 % of a C3 launch hold 43 % of the resident waves of the rest).  Sorting the end of the run puts the short tiles
// last; the tail of a run is the only place where that costs no locality worth having (list-scheduling simulation on the traced lifetimes: −1.5 … −3.4 % per XCD run).
__global__ void __launch_bounds__(1024) k_tile_order(const int* cost, const int* cscan, int ntile, int* order,
                                                     int* part, int nseg, XcdShares W, int keep_if_empty, int nclass, int tail_permille = 0, int tail_classes = 16) {
    // nseg contiguous segments per XCD, dealt round-robin (segment s of 8·nseg equal-cost segments goes to XCD s % 8):
    // with nseg = 1 an XCD's run is one stretch of the domain, and a stretch of interior fluid has no cheap tiles to
    // end its launch with.  The XCD's tiles are written to order[x·ntile …].
    __shared__ int s_min, s_max, s_wsum[16], s_off, s_beg[64], s_end[64];
    const int x = blockIdx.x;
    const long long total = cscan[ntile];
    if (total <= 0) {
        // keep_if_empty: a re-schedule from a sampled launch that never ran (its step was cancelled) keeps the old order
        if (!keep_if_empty && threadIdx.x == 0) { part[x] = 0; part[8 + x] = 0; }
        return;
    }
    auto lower_bound = [&](long long v) {
        int lo = 0, hi = ntile;                       // first t with cscan[t] >= v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cscan[mid] < v) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const int nall = 8 * nseg;
    if ((int)threadIdx.x < nseg) {                    // one binary search per segment bound, not per thread and loop
        // round k of the deal covers the cost span [k, k + 1) / nseg; inside it the XCDs take their shares in order
        const int k = (int)threadIdx.x, sg = x + 8 * k;
        const double lo = ((double)k + (double)W.cum[x]) / nseg, hi = ((double)k + (double)W.cum[x + 1]) / nseg;
        s_beg[threadIdx.x] = sg == 0 ? 0 : lower_bound((long long)((double)total * lo));
        s_end[threadIdx.x] = sg == nall - 1 ? ntile : lower_bound((long long)((double)total * hi));
    }
    __syncthreads();
    auto seg_beg = [&](int k) { return s_beg[k]; };
    auto seg_end = [&](int k) { return s_end[k]; };
    const int out0 = nseg == 1 ? seg_beg(0) : x * ntile;
    if (threadIdx.x == 0) { s_min = INT32_MAX; s_max = INT32_MIN; s_off = out0; }
    __syncthreads();
    int mn = INT32_MAX, mx = INT32_MIN;
    for (int k = 0; k < nseg; ++k) {
        const int beg = seg_beg(k), end = seg_end(k);
        for (int t = beg + (int)threadIdx.x; t < end; t += 1024) { const int c = cost[t]; if (c > 0) { mn = min(mn, c); mx = max(mx, c); } }
    }
    if (mn <= mx) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    __syncthreads();
    const int cmin = s_min;
    const float scale = s_max > cmin ? (float)nclass / (float)(s_max - cmin + 1) : 0.f;
    // class 0 = most expensive
    auto cls_of = [&](int c) { return nclass - 1 - min(nclass - 1, (int)((float)(c - cmin) * scale)); };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int cls = 0; cls < nclass; ++cls) {
        for (int k = 0; k < nseg; ++k) {
            const int beg = seg_beg(k), end = seg_end(k);
            for (int t0 = beg; t0 < end; t0 += 1024) {
                const int t = t0 + (int)threadIdx.x;
                const int c = t < end ? cost[t] : 0;
                const bool in = c > 0 && cls_of(c) == cls;                     // cost 0: the tile is not in this list
                const unsigned long long bal = __ballot(in);
                const int below = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) s_wsum[w] = __popcll(bal);
                __syncthreads();
                int woff = 0, tot = 0;
                for (int q = 0; q < 16; ++q) { const int v = s_wsum[q]; if (q < w) woff += v; tot += v; }
                const int base = s_off;
                if (in) order[base + woff + below] = t;
                __syncthreads();
                if (threadIdx.x == 0) s_off = base + tot;
                __syncthreads();
            }
        }
    }
    if (threadIdx.x == 0) { part[x] = out0; part[8 + x] = s_off - out0; }    // run start in order[], tiles in the run
    // ---- the end of the run once more, finer ------------------------------------------------------------------------------------------
    __shared__ int s_tail[2048], s_tcls[2048], s_tmin, s_tmax;
    __syncthreads();
    const int nrun = s_off - out0;
    int nt = tail_permille > 0 ? (int)((long long)nrun * tail_permille / 1000) : 0;
    nt = min(nt, 2048);
    if (nt < 64 || tail_classes < 2) return;
    const int tb = out0 + nrun - nt;
    if (threadIdx.x == 0) { s_tmin = INT32_MAX; s_tmax = INT32_MIN; }
    __syncthreads();
    for (int k = threadIdx.x; k < nt; k += 1024) {
        const int t = order[tb + k]; const int c = cost[t];
        s_tail[k] = t; s_tcls[k] = c;
        atomicMin(&s_tmin, c); atomicMax(&s_tmax, c);
    }
    __syncthreads();
    const int tmin = s_tmin;
    const float tscale = s_tmax > tmin ? (float)tail_classes / (float)(s_tmax - tmin + 1) : 0.f;
    for (int k = threadIdx.x; k < nt; k += 1024) s_tcls[k] = tail_classes - 1 - min(tail_classes - 1, (int)((float)(s_tcls[k] - tmin) * tscale));
    __syncthreads();
    // stable counting sort by class: one thread per class walks the ≤ 2 048 entries (sixteen classes, a few µs once per rebuild)
    __shared__ int s_cnt[64];
    if ((int)threadIdx.x < tail_classes && threadIdx.x < 64) {
        int n = 0;
        for (int k = 0; k < nt; ++k) n += s_tcls[k] == (int)threadIdx.x ? 1 : 0;
        s_cnt[threadIdx.x] = n;
    }
    __syncthreads();
    if ((int)threadIdx.x < tail_classes && threadIdx.x < 64) {
        int off = 0;
        for (int q = 0; q < (int)threadIdx.x; ++q) off += s_cnt[q];
        for (int k = 0; k < nt; ++k) if (s_tcls[k] == (int)threadIdx.x) order[tb + off++] = s_tail[k];
    }
}

// The tile schedule of a SMALL handle (≤ kSmallMaxTiles tiles, no slab) in ONE launch: k_tile_cost + the three scan launches +
// k_tile_order, every one of the eight workgroups (one per XCD run) repeating the costs and their scan in LDS.  `work`: the
// measured work of the tiles instead of the estimate (Engine::reschedule_from_work).  `bound`: no run may be longer — the host
// launches 8 × bound blocks until it has seen the table (it no longer waits for it), so the run boundaries are clamped to it.
constexpr int kSmallMaxTiles = 4096;
struct SmallSched {
    const int* key; const int* cstart; const int* work;
    int N, ntile, nxp, nxyp, D, ncell;
    int* order; int* part;
    XcdShares W;
    int nclass, bound, keep_if_empty;
    int* flags; StepCtrl* ctrl;      // block 0 hands a grid overflow of this rebuild to the step control (error 3) and re-arms the flag
};
__global__ void __launch_bounds__(1024) k_tile_schedule_small(const SmallSched S) {
    __shared__ int s_cost[kSmallMaxTiles], s_scan[kSmallMaxTiles + 1], s_w[16], s_b[9], s_min, s_max, s_cbase[16], s_cnt[16][16], s_off;
    const int x = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntile = S.ntile;
    if (x == 0 && tid == 0 && S.flags) {
        if (S.flags[kFlagOverflow]) { if (S.ctrl) S.ctrl->error = 3; S.flags[kFlagOverflowSeen] = 1; S.flags[kFlagOverflow] = 0; }
    }
    for (int t = tid; t < ntile; t += 1024) {
        int c;
        if (S.work) c = S.work[t];
        else {
            const int kf = S.key[t * 64], kl0 = S.key[min(t * 64 + 63, S.N - 1)];
            const int nseg = S.D == 3 ? 9 : 3;
            const int kl = min(kl0, S.ncell - 1);
            c = 64;
            for (int seg = 0; seg < nseg && kf < S.ncell; ++seg) {
                const int off = S.D == 3 ? ((seg % 3) - 1) * S.nxp + ((seg / 3) - 1) * S.nxyp : (seg - 1) * S.nxp;
                c += S.cstart[min(kl + off + 2, S.ncell)] - S.cstart[max(kf + off - 1, 0)];
            }
            if (kf >= S.ncell) c = 0;
        }
        s_cost[t] = c;
    }
    __syncthreads();
    int carry = 0;
    for (int b0 = 0; b0 < ntile; b0 += 1024) {
        const int idx = b0 + tid;
        const int v = idx < ntile ? s_cost[idx] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int t = s_w[k]; if (k < w) woff += t; tot += t; }
        if (idx < ntile) s_scan[idx] = carry + woff + inc - v;
        carry += tot;
        __syncthreads();
    }
    const long long total = carry;
    if (tid == 0) s_scan[ntile] = carry;
    if (total <= 0) {
        if (!S.keep_if_empty && tid == 0) { S.part[x] = 0; S.part[8 + x] = 0; }
        return;
    }
    __syncthreads();
    if (tid <= 8) {
        // run boundaries by cost share (first t with scan[t] >= share), as k_tile_order finds them
        const long long v = (long long)((double)total * (double)S.W.cum[tid]);
        int lo = 0, hi = ntile;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_scan[mid] < v) lo = mid + 1; else hi = mid; }
        s_b[tid] = tid == 0 ? 0 : (tid == 8 ? ntile : lo);
    }
    __syncthreads();
    if (tid == 0) {
        // no run longer than `bound` (8 × bound ≥ ntile): one forward pass
        int s = 0;
        for (int r = 0; r < 8; ++r) {
            int e = s_b[r + 1];
            e = min(e, s + S.bound); e = max(e, ntile - (7 - r) * S.bound); e = max(e, s); e = min(e, ntile);
            s_b[r] = s; s = e;
        }
        s_b[8] = ntile;
        s_min = INT32_MAX; s_max = INT32_MIN;
    }
    __syncthreads();
    const int beg = s_b[x], end = s_b[x + 1];
    int mn = INT32_MAX, mx = INT32_MIN;
    for (int t = beg + tid; t < end; t += 1024) { const int c = s_cost[t]; if (c > 0) { mn = min(mn, c); mx = max(mx, c); } }
    mn = wave_min_i(mn); mx = wave_max_i(mx);
    if (lane == 0 && mn <= mx) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    if (tid < 16) s_cbase[tid] = 0;
    __syncthreads();
    const int cmin = s_min, nclass = S.nclass;
    const float scale = s_max > cmin ? (float)nclass / (float)(s_max - cmin + 1) : 0.f;
    auto cls_of = [&](int c) { return nclass - 1 - min(nclass - 1, (int)((float)(c - cmin) * scale)); };
    // stable partition of the run into cost classes, most expensive first (k_tile_order walks the run once per class; here: the
    // class sizes first, then one pass that places every tile — three barriers per 1024 tiles instead of three per class)
    for (int t = beg + tid; t < end; t += 1024) { const int c = s_cost[t]; if (c > 0) atomicAdd(&s_cbase[cls_of(c)], 1); }
    __syncthreads();
    if (tid == 0) { int off = beg; for (int k = 0; k < 16; ++k) { const int n = s_cbase[k]; s_cbase[k] = off; off += n; } s_off = off; }
    __syncthreads();
    for (int t0 = beg; t0 < end; t0 += 1024) {
        const int t = t0 + tid;
        const int c = t < end ? s_cost[t] : 0;
        const int cls = c > 0 ? cls_of(c) : -1;
        int below = 0;
        for (int k = 0; k < nclass; ++k) {
            const unsigned long long bal = __ballot(cls == k);
            if (lane == 0) s_cnt[w][k] = __popcll(bal);
            if (cls == k) below = __popcll(bal & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (cls >= 0) {
            int off = s_cbase[cls];
            for (int q = 0; q < w; ++q) off += s_cnt[q][cls];
            S.order[off + below] = t;
        }
        __syncthreads();
        if (tid < nclass) { int tot = 0; for (int q = 0; q < 16; ++q) tot += s_cnt[q][tid]; s_cbase[tid] += tot; }
        __syncthreads();
    }
    if (tid == 0) { S.part[x] = beg; S.part[8 + x] = s_off - beg; }
}

__global__ void __launch_bounds__(256) k_scatter(int N, const int* key, const int* slot, const int* cstart,
                                                 int* tmp_idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) tmp_idx[cstart[key[i]] + slot[i]] = i;
}

// stable in-cell order: rank = #{cell mates whose previous index is smaller}.  The graveyard (dead particles:
// key == ncell, a whole ghost layer under domain decomposition) needs no order and keeps its scatter slots.
__global__ void __launch_bounds__(256) k_rankfix(int N, const int* key, const int* slot, const int* cstart, int ncell,
                                                 const int* tmp_idx, int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = key[i];
    const int s = cstart[k], e = cstart[k + 1];
    if (k == ncell) { perm[s + slot[i]] = i; return; }
    int rank = 0;
    for (int q = s; q < e; ++q) rank += tmp_idx[q] < i;
    perm[s + rank] = i;
}

// Domain decomposition: the reference's in-cell order is a history — a stable sort keeps cell mates in the order of
// their PREVIOUS sorted index (src/SPHCellList.jl:142), and that order decides which particle of a same-cell pair
// plays "i" (the density-diffusion term is not symmetric).  A slab's local index is not the global
// one (migrants arrive at the end of the array), so every particle carries an ORDER TAG that compares like its
// previous global sorted index: (global cell in sort order, rank in that cell) after a rebuild, the upload index
// before the first one.  Tags travel with migration and ghost-layer records.
__global__ void __launch_bounds__(256) k_rankfix_tag(int N, const int* key, const int* slot, const int* cstart, int ncell,
                                                     const int* tmp_idx, const unsigned long long* tag, int* perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = key[i];
    const int s = cstart[k], e = cstart[k + 1];
    if (k == ncell) { perm[s + slot[i]] = i; return; }
    const unsigned long long t = tag[i];
    int rank = 0;
    for (int q = s; q < e; ++q) {
        const int j = tmp_idx[q];
        const unsigned long long u = tag[j];
        rank += (u < t) | ((u == t) & (j < i));
    }
    perm[s + rank] = i;
}
// tags of the new order: 16 bits per global cell coordinate (z, y, x: the sort order of cells), 16 bits in-cell rank
__global__ void __launch_bounds__(256) k_make_tags(int N, const int* key, const int* cstart, GridDesc g, unsigned long long* tag) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int k = key[p];
    if (k >= g.ncell) { tag[p] = ~0ull; return; }
    const int cx = k % g.np[0], cy = (k / g.np[0]) % g.np[1], cz = k / (g.np[0] * g.np[1]);
    const unsigned long long X = (unsigned long long)(cx - 1 + g.gmin[0] + 32768), Y = (unsigned long long)(cy - 1 + g.gmin[1] + 32768),
                             Z = (unsigned long long)(cz - 1 + g.gmin[2] + 32768);
    tag[p] = (((Z << 16 | Y) << 16 | X) << 16) | (unsigned long long)((p - cstart[k]) & 0xFFFF);
}

template <class T>
struct PermuteArgs {
    using V4 = typename Vec4<T>::type;
    Half<const V4> pk0_in, pk1_in; const V4 *acc_in, *ghost_in;
    Half<V4> pk0_out, pk1_out; V4 *acc_out, *ghost_out;
    const uint8_t* type_in; uint8_t* type_out;
    const long long* id_in; long long* id_out;
    const unsigned long long* grp_in; unsigned long long* grp_out;
    const int* key_in; int* key_out;
    const unsigned long long* tag_in; unsigned long long* tag_out;     // order tags (domain decomposition only)
    const int* prow_in; int* prow_out;                                 // row at the last sphmi_download_permutation
    const V4* comp_in; V4* comp_out;                                   // low words of the double-float state (fp32 handles; else null)
    const int* perm;
    const int* flags;          // device-side rebuilds: flags[kFlagOverflow] set → copy instead of permuting (the order stays what it was)
    int* mdbc_list; int* mdbc_cnt;     // mDBC handles: the rows that carry a ghost node, appended in any order (*mdbc_cnt zeroed by k_cell_count)
    int N, has_ghost;
};

template <class T>
__global__ void __launch_bounds__(256) k_permute(const PermuteArgs<T> A) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.N) return;
    const int i = (A.flags && A.flags[kFlagOverflow]) ? p : A.perm[p];
    A.pk0_out[p] = A.pk0_in[i];
    A.pk1_out[p] = A.pk1_in[i];
    A.acc_out[p] = A.acc_in[i];
    if (A.has_ghost) {
        const auto gq = A.ghost_in[i];
        A.ghost_out[p] = gq;
        if (A.mdbc_list) {
            // one counter bump per wave: the waves' lists of ghost-bearing rows, in whatever order the waves arrive
            const bool has = gq.w != T(0);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(has);
            if (bal) {
                const int lane = threadIdx.x & 63, first = __builtin_ctzll(bal);
                int base = 0;
                if (lane == first) base = atomicAdd(A.mdbc_cnt, __builtin_popcountll(bal));
                base = __shfl(base, first, 64);
                if (has) A.mdbc_list[base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = p;
            }
        }
    }
    A.type_out[p] = A.type_in[i];
    A.id_out[p] = A.id_in[i];
    A.grp_out[p] = A.grp_in[i];
    A.key_out[p] = A.key_in[i];
    if (A.tag_in) A.tag_out[p] = A.tag_in[i];
    if (A.prow_in) A.prow_out[p] = A.prow_in[i];
    if (A.comp_in) A.comp_out[p] = A.comp_in[i];
}
__global__ void __launch_bounds__(256) k_iota(int* out, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = i;
}

// Pressure! (src/SimulationEquations.jl:18-24) on a state set: pk1.w = EOS(|pk0.w|)
template <class T>
__global__ void __launch_bounds__(256) k_eos(Half<const typename Vec4<T>::type> pk0, Half<typename Vec4<T>::type> pk1,
                                             int N, T rho0, T inv_rho0, T Cbe) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    auto v = pk1[i];
    v.w = eos7<T>(absT(pk0[i].w), rho0, inv_rho0, Cbe);
    pk1[i] = v;
}

// ---- start-up reductions (Δt on the uploaded state; Positionₙ⁺ = 0 → |x|) ---------------------
template <class T>
__global__ void __launch_bounds__(256) k_init_reduce(Half<const typename Vec4<T>::type> pk0,
                                                     Half<const typename Vec4<T>::type> pk1,
                                                     const typename Vec4<T>::type* acc, int N, T h, T eta2,
                                                     unsigned long long* red) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    T disp2 = 0, vis = 0, a2 = 0;
    if (i < N) {
        auto x = pk0[i]; auto v = pk1[i]; auto a = acc[i];
        const T rr = x.x * x.x + x.y * x.y + x.z * x.z;
        disp2 = rr;
        vis = absT(h * (v.x * x.x + v.y * x.y + v.z * x.z) / (rr + eta2));
        a2 = a.x * a.x + a.y * a.y + a.z * a.z;
    }
    disp2 = wave_max(disp2); vis = wave_max(vis); a2 = wave_max(a2);
    if ((threadIdx.x & 63) == 0) {
        atomic_max_bits(&red[0], disp2);
        atomic_max_bits(&red[1], vis);
        atomic_max_bits(&red[2], a2);
    }
}

// ---- mDBC -------------------------------------------------------------------------------------
template <class T> struct MdbcParams {
    using V4 = typename Vec4<T>::type;
    Half<V4> pk0;          // state A: ρ of boundary particles is rewritten in place
    V4* comp;              // fp32 handles: the low word of the rewritten ρ goes with it (ForceParams::comp); else null
    const V4* ghost;       // { g, flag }  flag != 0 ⇔ !iszero(GhostPoint)
    const uint8_t* type;   // ghost-copy bits (domain decomposition)
    const int* list; int n_list;   // the ghost-bearing particles (any order), rebuilt with the cell list by k_permute; null: every particle is looked at
    const int* cstart;
    GridDesc g;
    unsigned long long* red;   // red[3]: non-positive density flag
    int N;
    T H_inv;               // the cell hash stays in the handle's precision (same cells as the particles)
    double H2, h_inv, h, alphaD, m0, rho0, eta2;
    double gfac;           // αD·5 / (8h²): the Wendland gradient factor (src/SPHKernels.jl:85-86)
    int kernel;            // 0 WendlandC2, 1 CubicSpline
    const StepCtrl* ctrl;  // device-side step control (null: always run)
    // Control taken INSIDE this kernel (mDBC handles without moving bodies or slabs; the same scheme as ForceParams::ctl_in of
    // the predictor, which plain handles use): every wave takes the decisions of the step from the control block the previous
    // step left and the slots its corrector filled, block 0 stores the decided state to the OTHER block, where the predictor
    // and the corrector of this step read it.  `red` (the bad-density flag of this kernel) is then slot 3 of the set this
    // step's corrector fills: the PREVIOUS step's predictor has zeroed it (a zeroing by this launch would race with its own
    // blocks), and this step's predictor zeroes slots 0–2 of that set and slot 3 of the other one (ForceParams::mdbc_zero).
    const StepCtrl* ctl_in; StepCtrl* ctl_out;
    const unsigned long long* red_in;
    double ctl_h, ctl_c0, ctl_CFL;
};

template <class T> __device__ __forceinline__ T det3(T a00, T a01, T a02, T a10, T a11, T a12, T a20, T a21, T a22) {
    return a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
}

// One parked record against ghost node g: the exact test of src/SPHCellList.jl:336 and the moments of :337-359 (fp64 in both builds)
template <class T, int D>
__device__ __forceinline__ void mdbc_moments(const MdbcParams<T>& M, const double (&g)[3], const typename Vec4<T>::type n0,
                                             double (&b)[D + 1], double (&A)[D + 1][D + 1]) {
    constexpr int P = D + 1;
    using R = double;
    const R xij[3] = {g[0] - (R)n0.x, g[1] - (R)n0.y, D == 3 ? g[2] - (R)n0.z : R(0)};
    const R r2 = xij[0] * xij[0] + xij[1] * xij[1] + xij[2] * xij[2];
    if (!(r2 <= M.H2)) return;                                       // the exact test (src/SPHCellList.jl:336)
    // |xᵢⱼ| from the fp32 reciprocal square root + two Newton steps in fp64 (≈2⁻²⁴ → 2⁻⁴⁸ → below an ulp; the library's
    // correctly rounded sqrt is ≈25 instructions of a kernel that is bound by their number)
    R rr;
    {
        R y = (R)__builtin_amdgcn_rsqf((float)r2);
        y = y * (R(1.5) - R(0.5) * r2 * y * y);
        y = y * (R(1.5) - R(0.5) * r2 * y * y);
        rr = r2 > R(1e-30) ? r2 * y : sqrt(r2);                     // (the float conversion underflows below 1e-38)
    }
    R q = rr * M.h_inv;
    q = q > R(2) ? R(2) : q;
    const R tq = q - R(2);
    R Wij, fac;
    if (M.kernel == 1) {                                         // CubicSpline, src/SPHKernels.jl:89-106
        Wij = q <= R(1) ? M.alphaD * (R(1) - R(1.5) * q * q + R(0.75) * q * q * q) : M.alphaD * R(0.25) * (-(tq * tq * tq));
        const R dWdq = q <= R(1) ? M.alphaD * (R(-3) * q + R(2.25) * q * q) : M.alphaD * R(-0.75) * (tq * tq);
        fac = dWdq * M.h_inv / (rr + M.eta2);
    } else {
        const R t1 = R(1) - q * R(0.5);
        const R t2 = t1 * t1;
        Wij = M.alphaD * (t2 * t2) * (R(2) * q + R(1));          // src/SPHKernels.jl:75-78
        fac = M.gfac * (tq * tq * tq);                           // αD·5/(8h²)·(q − 2)³, :85-86
    }
    const R Vj = M.m0 * fast_rcp((R)n0.w);
    R fc[P];
    fc[0] = Vj * Wij;
    b[0] += M.m0 * Wij;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const R gw = fac * xij[d];
        fc[d + 1] = Vj * gw;
        b[d + 1] += M.m0 * gw;
    }
#pragma unroll
    for (int r = 0; r < P; ++r) {
        A[r][0] += fc[r];
#pragma unroll
        for (int k = 0; k < D; ++k) A[r][k + 1] += (-xij[k]) * fc[r];
    }
}

// ApplyMDBCCorrection (src/SPHCellList.jl:598-622) for particle i from the summed moments: one lane
template <class T, int D>
__device__ __forceinline__ void mdbc_apply(const MdbcParams<T>& M, const int i, const double (&g)[3], double (&b)[D + 1], double (&A)[D + 1][D + 1]) {
    constexpr int P = D + 1;
    using R = double;
    R det;
    if constexpr (P == 3) {
        det = det3<R>(A[0][0], A[0][1], A[0][2], A[1][0], A[1][1], A[1][2], A[2][0], A[2][1], A[2][2]);
    } else {
        det = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            R m[3][3];
            int cc = 0;
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                if (c2 == c) continue;
#pragma unroll
                for (int r = 1; r < 4; ++r) m[r - 1][cc] = A[r][c2];
                cc++;
            }
            const R d3 = det3<R>(m[0][0], m[0][1], m[0][2], m[1][0], m[1][1], m[1][2], m[2][0], m[2][1], m[2][2]);
            det += ((c & 1) ? R(-1) : R(1)) * A[0][c] * d3;
        }
    }
    auto me = M.pk0[i];
    R newrho = fabs((R)me.w);
    bool write = false;
    if (fabs(det) >= R(1e-3)) {
        // Gaussian elimination with partial pivoting on [A | b] (same order as the oracle)
        R Mx[P][P + 1], ipiv[P];
#pragma unroll
        for (int r = 0; r < P; ++r) {
#pragma unroll
            for (int c = 0; c < P; ++c) Mx[r][c] = A[r][c];
            Mx[r][P] = b[r];
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            int p = k;
            R best = fabs(Mx[k][k]);
#pragma unroll
            for (int r = k + 1; r < P; ++r) if (fabs(Mx[r][k]) > best) { best = fabs(Mx[r][k]); p = r; }
#pragma unroll
            for (int r = k + 1; r < P; ++r) {
                if (r == p) {
#pragma unroll
                    for (int c = 0; c <= P; ++c) { R t = Mx[k][c]; Mx[k][c] = Mx[r][c]; Mx[r][c] = t; }
                }
            }
            // (one reciprocal per pivot instead of a division per row — ten IEEE divisions of ≈15 instructions each on the one lane
            // the wave waits for; 1/pivot to 2⁻⁵², the system is solved to its conditioning either way)
            ipiv[k] = fast_rcp(Mx[k][k]);
#pragma unroll
            for (int r = k + 1; r < P; ++r) {
                const R f = Mx[r][k] * ipiv[k];
#pragma unroll
                for (int c = k; c <= P; ++c) Mx[r][c] -= f * Mx[k][c];
            }
        }
        R s[P];
#pragma unroll
        for (int r = P - 1; r >= 0; --r) {
            R acc = Mx[r][P];
#pragma unroll
            for (int c = r + 1; c < P; ++c) acc -= Mx[r][c] * s[c];
            s[r] = acc * ipiv[r];
        }
        const R xi[3] = {(R)me.x, (R)me.y, (R)me.z};
        R v1 = s[0];
#pragma unroll
        for (int d = 0; d < D; ++d) v1 += s[d + 1] * (xi[d] - g[d]);
        newrho = (v1 != v1) ? (R)M.rho0 : v1;
        write = true;
    } else if (A[0][0] > R(0)) {
        const R v = b[0] / A[0][0];
        newrho = (v != v) ? (R)M.rho0 : v;
        write = true;
    }
    if (write) {
        // the sign of ρ carries the MotionLimiter flag, so ρ must stay positive.  A ghost copy far out in a wide halo
        // sums over a truncated neighbourhood: whatever that gives is never read, must not flip the flag and is
        // no error (the owner of the particle judges the real value)
        if (!(newrho > R(0))) {
            if (M.type[i] & kGhostMask) return;
            atomicOr(&M.red[3], 1ull);
        }
        const T hi = (T)newrho;
        me.w = me.w > T(0) ? hi : -hi;
        M.pk0[i] = me;
        if (sizeof(T) == 4 && M.comp) M.comp[i].w = (T)(newrho - (R)hi);
    }
}

// All arithmetic of the moment matrix and its solve is fp64 in BOTH builds: the (D+1)×(D+1) systems of thin
// boundary layers are ill-conditioned, and fp32 accumulation flipped the |det| ≥ 1e-3 branch for a handful
// of particles of example/Dambreak2dMDBC.jl (density off by 3e-3); the kernel is a negligible part of a step.
//
// ONE WAVE per ghost node (round 4: per GHOST-BEARING particle — `list`, rebuilt with the cell list — not per particle: two
// thirds of the 54 820 waves of DucklingMDBC had nothing to do), in two phases, because what the round-2 counters showed was a
// LATENCY-bound kernel (vector ALU 38 % busy, ≈17 dependent load round trips per wave):
//   1. who is within H — the 3^(D-1) candidate rows of the node's cell neighbourhood in 64-candidate chunks, FOUR chunk loads in
//      flight per lane (nothing else is live yet: 16 registers), a cheap test in the handle's precision (a superset: the exact
//      fp64 test follows), and the records that pass are parked densely in LDS (ballot + prefix count);
//   2. the moments — every lane takes one parked record per round (two or three rounds instead of seventeen mostly-idle ones),
//      fp64 as before; the 64 partial sums of each of the (D+1)² + (D+1) moments are added in a FIXED order through LDS (one
//      half-fold with v_permlane32_swap, then lane v adds the 32 values of moment v: ≈150 instructions against ≈360 for a
//      shuffle tree over doubles); lane 0 solves.
template <class T, int D>
__global__ void __launch_bounds__(256) k_mdbc(const MdbcParams<T> M) {
    constexpr int P = D + 1, NV = P * P + P;
    constexpr int NROW = D == 3 ? 9 : 3;
    constexpr int U = sizeof(T) == 4 ? 4 : 2;              // chunk loads in flight (16 registers either way)
    constexpr int kStash = 64 * U + (sizeof(T) == 4 ? 128 : 64);      // records a wave can park: a full group of chunks on top of what is
                                                           // allowed to wait (24 KB per block either way); a typical node parks 100-170
    using R = double;
    using V4 = typename Vec4<T>::type;
    // (per wave: the stash of phase 1 / 2, and — once the last record has been worked off — the transpose buffer of the final sums)
    constexpr size_t kWaveLds = kStash * sizeof(V4) > NV * 33 * sizeof(double) ? kStash * sizeof(V4) : NV * 33 * sizeof(double);
    __shared__ __attribute__((aligned(16))) char s_lds[4][kWaveLds];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    bool have;
    int i;
    if (M.list) { have = wid < M.n_list; i = have ? M.list[wid] : 0; }
    else { have = wid < M.N; i = have ? wid : 0; if (have && M.ghost[i].w == T(0)) have = false; }
    if (M.ctl_in != nullptr) {
        // the decisions of the step, taken by the first wave of every block for its four (round 3: by every wave for itself — ≈100 fp64
        // instructions per ghost node, which is why handles above 32 768 particles kept the one-thread launch); block 0 stores them
        __shared__ int s_active;
        if (threadIdx.x < 64) {
            StepCtrl c = *M.ctl_in;
            const unsigned long long r0 = M.red_in[0], r1 = M.red_in[1], r2 = M.red_in[2], r3 = M.red_in[3];
            (void)step_control_decide<T>(r0, r1, r2, r3, c, M.ctl_h, M.ctl_c0, M.ctl_CFL);
            if (blockIdx.x == 0 && threadIdx.x == 0) *M.ctl_out = c;
            if (threadIdx.x == 0) s_active = c.active;
        }
        __syncthreads();
        if (!s_active) return;
    } else if (M.ctrl && !M.ctrl->active) return;
    if (!have) return;
    const auto gq = M.ghost[i];
    const T gT[3] = {gq.x, gq.y, gq.z};
    const R g[3] = {(R)gq.x, (R)gq.y, (R)gq.z};
    int gc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) gc[d] = d < D ? map_floor<T>(gT[d], M.H_inv) - M.g.gmin[d] + 1 : 0;
    // row r of the neighbourhood: the three x-adjacent cells are one contiguous range, clipped to the padded grid (lane r < NROW)
    int rs = 0, rc = 0;
    if (lane < NROW) {
        const int sy = lane % 3, sz = lane / 3;
        const int cy = gc[1] + sy - 1, cz = D == 3 ? gc[2] + sz - 1 : 0;
        int x0 = gc[0] - 1, x1 = gc[0] + 1;
        if (x0 < 0) x0 = 0;
        if (x1 > M.g.np[0] - 1) x1 = M.g.np[0] - 1;
        if (cy >= 0 && cy < M.g.np[1] && cz >= 0 && cz < M.g.np[2] && x0 <= x1) {
            const int row = M.g.np[0] * (cy + M.g.np[1] * cz);
            rs = M.cstart[row + x0];
            rc = M.cstart[row + x1 + 1] - rs;
        }
    }
    // the chunks ("jobs") of all rows, numbered through: row r holds jobs [jb[r], jb[r + 1])
    int jincl = (rc + 63) >> 6;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(jincl, o, 64); if (lane >= o) jincl += u; }
    const int njobs = __builtin_amdgcn_readlane(jincl, NROW - 1);
    // job table: lane l describes job l of the current block of 64 jobs — first record and records left in its row — so that the
    // loop below fetches a job's description with two v_readlane (a row look-up per chunk was a chain of 58 vector selects)
    const int jexcl = jincl - ((rc + 63) >> 6);
    int job_first = 0, job_left = 0;
    auto describe_jobs = [&](const int jbase) __attribute__((always_inline)) {
        const int jb = jbase + lane;
        job_first = 0; job_left = 0;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int e = __builtin_amdgcn_readlane(jexcl, r), st = __builtin_amdgcn_readlane(rs, r), cn = __builtin_amdgcn_readlane(rc, r);
            const int o = (jb - e) << 6;
            if (jb >= e && o < cn) { job_first = st + o; job_left = cn - o; }
        }
    };
    R b[P], A[P][P];   // A[r][c]
#pragma unroll
    for (int r = 0; r < P; ++r) { b[r] = 0;
#pragma unroll
        for (int c = 0; c < P; ++c) A[r][c] = 0; }
    V4* const stash = reinterpret_cast<V4*>(s_lds[wv]);
    const T H2t = sizeof(T) == 4 ? (T)(M.H2 * (1.0 + 1e-5)) : (T)M.H2;       // fp32: a superset of what the fp64 test accepts
    int job0 = 0, parked = 0;
    do {
        // phase 1: U chunk loads in flight, test, park — until the rows are through or the stash could not take another group
        int npend = 0;
        for (; job0 < njobs && npend <= kStash - 64 * U; job0 += U) {
            if ((job0 & 63) == 0) describe_jobs(job0);                         // (U divides 64: a group never straddles two blocks of jobs)
            V4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jl = __builtin_amdgcn_readfirstlane((job0 & 63) + u);
                const int first = __builtin_amdgcn_readlane(job_first, jl), left = __builtin_amdgcn_readlane(job_left, jl);
                ok[u] = lane < left;                                            // (a job beyond the last one has nothing left)
                v[u] = M.pk0[ok[u] ? first + lane : i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const T dx = gT[0] - v[u].x, dy = gT[1] - v[u].y, dz = D == 3 ? gT[2] - v[u].z : T(0);
                const bool pass = ok[u] && v[u].w > T(0) && (dx * dx + dy * dy + dz * dz) <= H2t;       // ParticleType[j] == Fluid, within H
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
                if (pass) stash[npend + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = v[u];
                npend += __builtin_popcountll(bal);
            }
        }
        // phase 2 on the parked records
        const int n = npend;
        parked += n;
        wave_sync();
        for (int k0 = 0; k0 < n; k0 += 64) {
            if (k0 + lane >= n) continue;
            const V4 n0 = stash[k0 + lane];
            mdbc_moments<T, D>(M, g, n0, b, A);
        }
        wave_sync();
    } while (job0 < njobs);
    // no Fluid record near the node (a dry wall: a third of DucklingMDBC's nodes): all moments are zero, the particle stays as it is (:618-621)
    if (parked == 0) return;
    // the 64 partial sums of every moment, added in a fixed order: upper half onto lower half, then lane v adds row v
    {
        double (*red)[33] = reinterpret_cast<double (*)[33]>(s_lds[wv]);
        auto fold = [&](R x) -> R {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
            unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32), lo2 = lo, hi2 = hi;
            swap_halves(lo, lo2); swap_halves(hi, hi2);      // lo2 / hi2: lanes 0-31 now hold the words of lanes 32-63
            return x + __longlong_as_double((long long)(((unsigned long long)hi2 << 32) | lo2));
        };
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const R fb = fold(b[r]);
            if (lane < 32) red[r][lane] = fb;
#pragma unroll
            for (int c = 0; c < P; ++c) { const R fa = fold(A[r][c]); if (lane < 32) red[P + r * P + c][lane] = fa; }
        }
        wave_sync();
        if (lane < NV) {
            R s = 0;
            for (int k = 0; k < 32; ++k) s += red[lane][k];
            red[lane][32] = s;
        }
        wave_sync();
        if (lane != 0) return;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            b[r] = red[r][32];
#pragma unroll
            for (int c = 0; c < P; ++c) A[r][c] = red[P + r * P + c][32];
        }
    }
    mdbc_apply<T, D>(M, i, g, b, A);
}

template <int CTRL> __device__ __forceinline__ double dpp_f64(const double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// the sum over the 16 lanes of a DPP row, in every lane of the row, bit-identical in all of them (each step adds the same two
// values in both lanes of a pair): lane ^ 1, lane ^ 2, then the mirror images inside 8 and inside 16 lanes
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_f64<0xB1>(x);        // quad_perm [1, 0, 3, 2]
    x += dpp_f64<0x4E>(x);        // quad_perm [2, 3, 0, 1]
    x += dpp_f64<0x141>(x);       // row_half_mirror
    x += dpp_f64<0x140>(x);       // row_mirror
    return x;
}

// SIXTEEN LANES per ghost node, four nodes per wave — the launch for handles with thousands of nodes and sparse neighbourhoods
// (DucklingMDBC: 226 candidate records and 9 accepted ones per node, a third of the nodes next to no Fluid at all).  With one
// wave per node the candidate rows of such a case fill 25 of the 64 lanes of a chunk, and the per-wave fixed costs — the row
// look-ups, the reduction of the (D+1)² + (D+1) moments, the solve on one lane — are paid 21 408 times.  Here they are paid once per
// four nodes: a node's rows go through in chunks of 16, the records that pass are parked per group, every lane of a group takes one
// parked record per round, the 16 partial sums of a moment are added inside the DPP row, and lane 0 of each group solves.  Same
// arithmetic per record as k_mdbc (mdbc_moments / mdbc_apply); only the order of the sums differs.  Needs the node list.
// Registers: 128 (four waves per SIMD; the 20 fp64 sums of a 3-D node are 40 of them).  Builds held to 96 / 80 registers spill 136 / 216
// bytes and run DucklingMDBC's launch in 32 / 66 µs against 27 (184 registers, two waves per SIMD: 29).
template <class T, int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_mdbc_group(const MdbcParams<T> M) {
    constexpr int P = D + 1;
    constexpr int G = 16, NG = 4;
    constexpr int NROW = D == 3 ? 9 : 3;
    constexpr int U = sizeof(T) == 4 ? 4 : 2;
    constexpr int kStashG = G * U + (sizeof(T) == 4 ? 32 : 16);      // records a group can park (24 KB per block either way)
    using R = double;
    using V4 = typename Vec4<T>::type;
    __shared__ __attribute__((aligned(16))) V4 s_stash[4][NG][kStashG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, grp = lane >> 4, sub = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int slot = wid * NG + grp;
    const bool have = slot < M.n_list;
    int i = have ? M.list[slot] : 0;
    if (M.ctl_in != nullptr) {
        __shared__ int s_active;
        if (threadIdx.x < 64) {
            StepCtrl c = *M.ctl_in;
            const unsigned long long r0 = M.red_in[0], r1 = M.red_in[1], r2 = M.red_in[2], r3 = M.red_in[3];
            (void)step_control_decide<T>(r0, r1, r2, r3, c, M.ctl_h, M.ctl_c0, M.ctl_CFL);
            if (blockIdx.x == 0 && threadIdx.x == 0) *M.ctl_out = c;
            if (threadIdx.x == 0) s_active = c.active;
        }
        __syncthreads();
        if (!s_active) return;
    } else if (M.ctrl && !M.ctrl->active) return;
    if (__builtin_amdgcn_ballot_w64(have) == 0) return;
    const auto gq = M.ghost[i];
    const T gT[3] = {gq.x, gq.y, gq.z};
    int gc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) gc[d] = d < D ? map_floor<T>(gT[d], M.H_inv) - M.g.gmin[d] + 1 : 0;
    int rs = 0, rc = 0;
    if (have && sub < NROW) {
        const int sy = sub % 3, sz = sub / 3;
        const int cy = gc[1] + sy - 1, cz = D == 3 ? gc[2] + sz - 1 : 0;
        int x0 = gc[0] - 1, x1 = gc[0] + 1;
        if (x0 < 0) x0 = 0;
        if (x1 > M.g.np[0] - 1) x1 = M.g.np[0] - 1;
        if (cy >= 0 && cy < M.g.np[1] && cz >= 0 && cz < M.g.np[2] && x0 <= x1) {
            const int row = M.g.np[0] * (cy + M.g.np[1] * cz);
            rs = M.cstart[row + x0];
            rc = M.cstart[row + x1 + 1] - rs;
        }
    }
    // the chunks ("jobs") of a node's rows, numbered through per group; lane `sub` of a group describes job jbase + sub of its node
    const int jrow = (rc + G - 1) / G;
    int jincl = jrow;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(jincl, o, G); if (sub >= o) jincl += u; }
    const int njobs = __shfl(jincl, NROW - 1, G);
    const int jexcl = jincl - jrow;
    int job_first = 0, job_left = 0;
    auto describe_jobs = [&](const int jbase) __attribute__((always_inline)) {
        const int jb = jbase + sub;
        job_first = 0; job_left = 0;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int e = __shfl(jexcl, r, G), st = __shfl(rs, r, G), cn = __shfl(rc, r, G);
            const int o = (jb - e) * G;
            if (jb >= e && o < cn) { job_first = st + o; job_left = cn - o; }
        }
    };
    R b[P], A[P][P];
#pragma unroll
    for (int r = 0; r < P; ++r) { b[r] = 0;
#pragma unroll
        for (int c = 0; c < P; ++c) A[r][c] = 0; }
    V4* const stash = s_stash[wv][grp];
    const T H2t = sizeof(T) == 4 ? (T)(M.H2 * (1.0 + 1e-5)) : (T)M.H2;
    const unsigned below = (1u << sub) - 1u;
    int job0 = 0, parked = 0;
    int npend = 0;
    do {
        while (__builtin_amdgcn_ballot_w64(job0 < njobs) != 0 && __builtin_amdgcn_ballot_w64(npend > kStashG - G * U) == 0) {
            if ((job0 & (G - 1)) == 0) describe_jobs(job0);
            V4 v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jl = (job0 & (G - 1)) + u;
                const int first = __shfl(job_first, jl, G), left = __shfl(job_left, jl, G);
                ok[u] = sub < left;
                v[u] = M.pk0[ok[u] ? first + sub : 0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const T dx = gT[0] - v[u].x, dy = gT[1] - v[u].y, dz = D == 3 ? gT[2] - v[u].z : T(0);
                const bool pass = ok[u] && v[u].w > T(0) && (dx * dx + dy * dy + dz * dz) <= H2t;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
                const unsigned w = lane < 32 ? (unsigned)bal : (unsigned)(bal >> 32);
                const unsigned gb = (w >> (lane & 16)) & 0xFFFFu;                 // the 16 verdicts of my group
                if (pass) stash[npend + __builtin_popcount(gb & below)] = v[u];
                npend += __builtin_popcount(gb);
            }
            job0 += U;
        }
        // phase 2 for the groups whose stash is full, for all of them once every row is through: which lane takes which record of a
        // node — the order of its sums — then depends on that node's own rows only, not on the three nodes it shares the wave with
        // (the list is appended by atomics: its order differs from run to run)
        const bool last = __builtin_amdgcn_ballot_w64(job0 < njobs) == 0;
        const int n = (last || npend > kStashG - G * U) ? npend : 0;
        parked |= npend;
        wave_sync();
        for (int k0 = 0; __builtin_amdgcn_ballot_w64(k0 < n) != 0; k0 += G) {
            if (k0 + sub < n) {
                const V4 n0 = stash[k0 + sub];
                const R g[3] = {(R)gT[0], (R)gT[1], (R)gT[2]};
                mdbc_moments<T, D>(M, g, n0, b, A);
            }
        }
        if (n != 0) npend = 0;
        wave_sync();
        if (last) break;
    } while (true);
    if (__builtin_amdgcn_ballot_w64(parked != 0) == 0) return;          // four dry nodes
#pragma unroll
    for (int r = 0; r < P; ++r) {
        b[r] = row16_sum(b[r]);
#pragma unroll
        for (int c = 0; c < P; ++c) A[r][c] = row16_sum(A[r][c]);
    }
    // (slot and node index are worked out again from the lane number the hardware counts, behind an asm — a value the compiler cannot merge with the
    // ones above and therefore does not keep, or spill, through the loops: the kernel has not one register to spare)
    int lane_now;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
    const int slot_now = wid * NG + (lane_now >> 4);
    if ((lane_now & 15) != 0 || slot_now >= M.n_list || parked == 0) return;
    i = M.list[slot_now];
    const R g[3] = {(R)gT[0], (R)gT[1], (R)gT[2]};
    mdbc_apply<T, D>(M, i, g, b, A);
}

// Output side: the SimParticles fields in the HOST's layout and float type, packed on the device
// so that sphmi_download is one kernel + one copy per field (the host loops of the first version cost more than
// the 25 steps between two outputs of the 1 M-particle case).
template <class H> struct OutFields {
    H *pos, *vel, *acc, *rho, *press, *ghost;      // N×D, N×D, N×D, N, N, N×D  (null = not requested)
    long long* cells;                              // N×D
};
template <class T, class H>
__global__ void __launch_bounds__(256) k_pack_output(Half<const typename Vec4<T>::type> pk0, Half<const typename Vec4<T>::type> pk1,
                                                     Half<const typename Vec4<T>::type> half0, const typename Vec4<T>::type* accv,
                                                     const typename Vec4<T>::type* ghostv, const typename Vec4<T>::type* comp,
                                                     const int* key, int N, int D, int C,
                                                     GridDesc g, int have_grid, T rho0, T inv_rho0, T Cbe, OutFields<H> o) {
    // C: components per vector in the output (D, or 3 for the VTKHDF point layout: 2-D handles keep z = 0)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const size_t b = (size_t)i * C;
    if (o.pos || o.rho) {
        const auto q = pk0[i];
        if (sizeof(T) == 4 && comp) {
            // fp32 handles: the state is the record + its low word (ForceParams::comp) — a Float64 caller gets both
            const auto c = comp[i];
            if (o.pos) { o.pos[b] = (H)((double)q.x + (double)c.x); o.pos[b + 1] = (H)((double)q.y + (double)c.y); if (C == 3) o.pos[b + 2] = (H)((double)q.z + (double)c.z); }
            if (o.rho) o.rho[i] = (H)((double)(q.w < T(0) ? -q.w : q.w) + (double)c.w);
        } else {
            if (o.pos) { o.pos[b] = (H)q.x; o.pos[b + 1] = (H)q.y; if (C == 3) o.pos[b + 2] = (H)q.z; }
            if (o.rho) o.rho[i] = (H)(q.w < T(0) ? -q.w : q.w);
        }
    }
    if (o.vel || (o.press && !half0)) {
        const auto q = pk1[i];
        if (o.vel) { o.vel[b] = (H)q.x; o.vel[b + 1] = (H)q.y; if (C == 3) o.vel[b + 2] = (H)q.z; }
        if (o.press && !half0) o.press[i] = (H)q.w;                 // before any step: Pressure!(ρ) of :835
    }
    if (o.press && half0) {
        // SimParticles.Pressure holds Pressure!(ρₙ⁺) of the last step (src/SPHCellList.jl:789)
        const T w = half0[i].w;
        const T rr = sizeof(T) == 8 ? w / rho0 : w * inv_rho0;
        const T r2 = rr * rr, r4 = r2 * r2;
        o.press[i] = (H)(Cbe * (r4 * r2 * rr - T(1)));
    }
    if (o.acc) { const auto q = accv[i]; o.acc[b] = (H)q.x; o.acc[b + 1] = (H)q.y; if (C == 3) o.acc[b + 2] = (H)q.z; }
    if (o.ghost) { const auto q = ghostv[i]; o.ghost[b] = (H)q.x; o.ghost[b + 1] = (H)q.y; if (C == 3) o.ghost[b + 2] = (H)q.z; }
    if (o.cells) {
        const size_t bc = (size_t)i * D;          // CartesianIndex{D}: never padded
        if (!have_grid) { o.cells[bc] = 0; o.cells[bc + 1] = 0; if (D == 3) o.cells[bc + 2] = 0; }
        else {
            int kk = key[i];
            const int cx = kk % g.np[0]; kk /= g.np[0];
            const int cy = kk % g.np[1], cz = kk / g.np[1];
            o.cells[bc] = (long long)cx - 1 + g.gmin[0];
            o.cells[bc + 1] = (long long)cy - 1 + g.gmin[1];
            if (D == 3) o.cells[bc + 2] = (long long)cz - 1 + g.gmin[2];
        }
    }
}

// ---- pre-processing on the device: the 3-D dam-break lattice of SURVEY §8d generated where it is used ----
// The reference reads its particle layouts from CSV files written by DualSPHysics' GenCase (src/PreProcess.jl:45-119) and ships
// the 3-D dam break only at Dp 0.02 (the Dp0.0085 / Dp0.005 blobs are missing); the bench resolutions are generated.  Same
// lattice, same order, same IDs and densities as the host generator (sphexample_amd/cases.py: dam_break_3d_arrays), which
// reproduces the shipped Dp0.02 files: node (i, j, k) ↦ dp/2 + dp·(i, j, k), boundary first (tank, then the pillar object;
// x-major / z-fastest inside each), fluid block after it with the hydrostatic density of the inverse EOS.
struct DamBreakGrid {
    int nx, ny, nk;                 // boundary lattice extents (k = 0 … kmax)
    int kwall, kcap, pi0, pi1, pj0, pj1;
    int fi, fj, fk;                 // fluid block extents (indices 1 … f*)
    double dp, rho0, g, B;
};
// site class of boundary lattice node (i, j, k): 1 = tank, 2 = pillar object, 0 = none
__device__ __forceinline__ int dam_break_site(const DamBreakGrid& G, int i, int j, int k) {
    const bool tank_perim = i == 0 || i == G.nx - 1 || j == 0 || j == G.ny - 1;
    const bool in_pillar = i >= G.pi0 && i <= G.pi1 && j >= G.pj0 && j <= G.pj1;
    const bool pillar_int = i > G.pi0 && i < G.pi1 && j > G.pj0 && j < G.pj1;
    const bool bottom = k == 0 && !pillar_int;
    const bool walls = k >= 1 && k <= G.kwall && tank_perim;
    const bool shell = k >= 1 && k <= G.kcap - 1 && in_pillar && !pillar_int;
    const bool cap = k == G.kcap && in_pillar;
    if ((bottom && !in_pillar) || walls) return 1;
    if ((bottom && in_pillar) || shell || cap) return 2;
    return 0;
}
__global__ void __launch_bounds__(256) k_gen_flags(DamBreakGrid G, long long M, int which, int* flag) {
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M) return;
    const int k = (int)(s % G.nk), j = (int)((s / G.nk) % G.ny), i = (int)(s / ((long long)G.nk * G.ny));
    flag[s] = dam_break_site(G, i, j, k) == which ? 1 : 0;
}
template <class T>
__global__ void __launch_bounds__(256) k_gen_boundary(DamBreakGrid G, long long M, const int* flag, const int* pos, int base,
                                                      Half<typename Vec4<T>::type> pk0, Half<typename Vec4<T>::type> pk1, uint8_t* type,
                                                      long long* id, unsigned long long* grp, typename Vec4<T>::type* comp) {
#pragma clang fp contract(off)        // o + dp·i rounded twice, as the host generator (and GenCase's files) have it
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M || !flag[s]) return;
    const int k = (int)(s % G.nk), j = (int)((s / G.nk) % G.ny), i = (int)(s / ((long long)G.nk * G.ny));
    const int p = base + pos[s];
    typename Vec4<T>::type q0, q1;
    const double o = G.dp / 2;
    const double xd = o + G.dp * (double)i, yd = o + G.dp * (double)j, zd = o + G.dp * (double)k;
    q0.x = (T)xd; q0.y = (T)yd; q0.z = (T)zd;
    q0.w = -(T)G.rho0;                                  // Fixed: MotionLimiter 0 → negative sign
    q1.x = q1.y = q1.z = q1.w = T(0);
    pk0[p] = q0; pk1[p] = q1; type[p] = 2; id[p] = p + 1; grp[p] = 1;
    if (comp) {                                         // what the Float64 layout holds beyond the fp32 record (sphmi_upload does the same)
        typename Vec4<T>::type c;
        c.x = (T)(xd - (double)q0.x); c.y = (T)(yd - (double)q0.y); c.z = (T)(zd - (double)q0.z); c.w = (T)(G.rho0 - (double)(T)G.rho0);
        comp[p] = c;
    }
}
template <class T>
__global__ void __launch_bounds__(256) k_gen_fluid(DamBreakGrid G, int nb, int nf, Half<typename Vec4<T>::type> pk0,
                                                   Half<typename Vec4<T>::type> pk1, uint8_t* type, long long* id, unsigned long long* grp,
                                                   typename Vec4<T>::type* comp) {
#pragma clang fp contract(off)
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const int k = f % G.fk + 1, j = (f / G.fk) % G.fj + 1, i = f / (G.fk * G.fj) + 1;
    const double o = G.dp / 2, z = o + G.dp * (double)k, ztop = o + G.dp * (double)G.fk;
    typename Vec4<T>::type q0, q1;
    const double xd = o + G.dp * (double)i, yd = o + G.dp * (double)j, rd = G.rho0 * pow(1.0 + G.rho0 * G.g * (ztop - z) / G.B, 1.0 / 7.0);
    q0.x = (T)xd; q0.y = (T)yd; q0.z = (T)z;
    q0.w = (T)rd;
    q1.x = q1.y = q1.z = q1.w = T(0);
    const int p = nb + f;
    pk0[p] = q0; pk1[p] = q1; type[p] = 1; id[p] = p + 1; grp[p] = 2;
    if (comp) {
        typename Vec4<T>::type c;
        c.x = (T)(xd - (double)q0.x); c.y = (T)(yd - (double)q0.y); c.z = (T)(z - (double)q0.z); c.w = (T)(rd - (double)q0.w);
        comp[p] = c;
    }
}

// UniqueCells (src/SPHCellList.jl:148-157) on the device: heads of the runs of equal keys → compacted cell coordinates
__global__ void __launch_bounds__(256) k_cell_heads(const int* key, int N, int ncell, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { const int k = key[i]; flag[i] = (k < ncell && (i == 0 || key[i - 1] != k)) ? 1 : 0; }
}
__global__ void __launch_bounds__(256) k_cells_out(const int* key, const int* flag, const int* pos, int N, GridDesc g, int D, long long* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !flag[i]) return;
    int kk = key[i];
    const int cx = kk % g.np[0]; kk /= g.np[0];
    const int cy = kk % g.np[1], cz = kk / g.np[1];
    long long* o = out + (size_t)pos[i] * D;
    o[0] = (long long)cx - 1 + g.gmin[0]; o[1] = (long long)cy - 1 + g.gmin[1];
    if (D == 3) o[2] = (long long)cz - 1 + g.gmin[2];
}

// ProgressMotion, src/SPHCellList.jl:575-596: particles of Type Moving whose GroupMarker has a MotionDetails get
// Velocity = v·dir·ShouldMove and Position += Velocity·dt/2 (state set A, in place).
struct MotionTable {
    int n;
    unsigned long long group[16];
    double vel[16], start[16], dur[16], dir[16][3];
};
// the step control taken inside the first k_progress_motion of a step (the scheme of ForceParams::ctl_in); ctl_in = null: not
struct MotionCtl { const StepCtrl* ctl_in; StepCtrl* ctl_out; const unsigned long long* red_in; unsigned long long* red_zero; double h, c0, CFL; };
template <class T>
__global__ void __launch_bounds__(256) k_progress_motion(Half<typename Vec4<T>::type> pk0, Half<typename Vec4<T>::type> pk1,
                                                         const uint8_t* type, const unsigned long long* group, int N,
                                                         MotionTable M, double total_time, double dt2, const StepCtrl* ctrl,
                                                         typename Vec4<T>::type* comp, const MotionCtl mc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = i < N && (type[i] & 0x3F) == 3;
    if (mc.ctl_in != nullptr) {
        // the control of the step taken HERE (handles with moving bodies and nothing else in front of their predictor): the Moving
        // particles need the decisions — Δt/2, the clock — and the first thread stores them for the kernels behind
        if (!mine && i != 0) return;
        StepCtrl c = *mc.ctl_in;
        const unsigned long long r0 = mc.red_in[0], r1 = mc.red_in[1], r2 = mc.red_in[2], r3 = mc.red_in[3];
        const bool consumed = step_control_decide<T>(r0, r1, r2, r3, c, mc.h, mc.c0, mc.CFL);
        if (i == 0) {
            *mc.ctl_out = c;
            if (consumed) { mc.red_zero[0] = 0; mc.red_zero[1] = 0; mc.red_zero[2] = 0; mc.red_zero[3] = 0; }
        }
        if (!c.active) return;
        total_time = c.t_step_start; dt2 = c.dt2;
    } else if (ctrl) { if (!ctrl->active) return; total_time = ctrl->t_step_start; dt2 = ctrl->dt2; }
    if (!mine) return;
    const unsigned long long g = group[i];
    for (int m = 0; m < M.n; ++m) {
        if (M.group[m] != g) continue;
        const double on = (M.start[m] <= total_time && total_time <= M.start[m] + M.dur[m]) ? 1.0 : 0.0;
        auto q0 = pk0[i]; auto q1 = pk1[i];
        const T vx = (T)(M.vel[m] * M.dir[m][0] * on), vy = (T)(M.vel[m] * M.dir[m][1] * on), vz = (T)(M.vel[m] * M.dir[m][2] * on);
        q1.x = vx; q1.y = vy; q1.z = vz;
        if (sizeof(T) == 4 && comp) {
            // a prescribed motion is integrated twice per step, in place: as a double-float like the corrector's (ForceParams::comp)
            auto c = comp[i];
            const double xd = ((double)q0.x + (double)c.x) + (double)vx * dt2, yd = ((double)q0.y + (double)c.y) + (double)vy * dt2,
                         zd = ((double)q0.z + (double)c.z) + (double)vz * dt2;
            q0.x = (T)xd; q0.y = (T)yd; q0.z = (T)zd;
            c.x = (T)(xd - (double)q0.x); c.y = (T)(yd - (double)q0.y); c.z = (T)(zd - (double)q0.z);
            comp[i] = c;
        } else { q0.x += vx * (T)dt2; q0.y += vy * (T)dt2; q0.z += vz * (T)dt2; }
        pk0[i] = q0; pk1[i] = q1;
        return;
    }
}

// ---- domain decomposition (one process per GPU, x-slabs; csrc/sphmi_multi.h) -----------
// global cell index along the slab axis of every particle, current order
template <class T>
__global__ void __launch_bounds__(256) k_dd_cellx(Half<const typename Vec4<T>::type> pk0, int N, T inv_cutoff, int axis, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        const typename Vec4<T>::type q = pk0[i];
        out[i] = map_floor<T>(axis == 0 ? q.x : (axis == 1 ? q.y : q.z), inv_cutoff);
    }
}

// Load balance by WORK, not by particle count: the cost of a particle in the neighbour kernel goes with the candidates of
// its 3^D cells (a dry wall particle has a tenth of an interior fluid particle's).  Per cell column along the slab axis:
// Σ over OWNED particles of that candidate count, from the cell list of the last rebuild.
// (One device-scope 64-bit atomic per particle on a few hundred counters was 2.6 ms at 0.5 M particles and 5 ms at 1 M — round 4's
// trace of two slabs on one GPU, profiles/r04_slab_overhead.md: the lanes of a wave hold a handful of columns, so equal neighbours
// are merged by a segmented scan first — one atomic per run of equal columns in a wave, ≈60 µs.)
__global__ void __launch_bounds__(256) k_dd_column_cost(const int* key, const uint8_t* type, const int* cstart, int N, GridDesc g,
                                                        int D, int axis, long long col0, int ncols, unsigned long long* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long long col = -1;
    int c = 0;
    if (i < N) {
        const uint8_t t = type[i];
        const int k = key[i];
        if (t != 0 && !(t & kGhostMask) && k < g.ncell) {
            const int nxp = g.np[0], nxyp = g.np[0] * g.np[1];
            const int nseg = D == 3 ? 9 : 3;
            for (int seg = 0; seg < nseg; ++seg) {
                const int off = D == 3 ? ((seg % 3) - 1) * nxp + ((seg / 3) - 1) * nxyp : (seg - 1) * nxp;
                c += cstart[k + off + 2] - cstart[k + off - 1];
            }
            const int cc[3] = {k % nxp, (k / nxp) % g.np[1], k / nxyp};
            col = (long long)cc[axis] - 1 + g.gmin[axis] - col0;
            if (col < 0 || col >= ncols) col = -1;
        }
    }
    // runs of equal columns inside the wave (the particles are cell-sorted): the head of a run adds the run's sum
    const int icol = (int)col;
    const int prev = __shfl_up(icol, 1, 64);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(lane == 0 || icol != prev);
    unsigned long long sum = (unsigned long long)c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long u = __shfl_down(sum, o, 64);
        // lanes lane+1 … lane+o belong to my run iff no head sits in (lane, lane+o]
        const unsigned long long between = lane + o < 64 ? (heads >> (lane + 1)) & ((o >= 64 ? ~0ull : ((1ull << o) - 1ull))) : 1ull;
        if (lane + o < 64 && between == 0) sum += u;
    }
    if (((heads >> lane) & 1ull) && icol >= 0 && sum) atomicAdd(&out[icol], sum);
}

// Migration record buffer for n particles:
// [n×V4 pk0][n×V4 pk1][n×V4 acc][n×V4 mDBC ghost node][fp32 records only: n×V4 low words of the state][n×i64 id][n×u64 group][n×u64 order tag][n×i32 row at the last
// sphmi_download_permutation, padded to 8 bytes][n×u8 type, padded to 8 bytes]
template <class T> struct DdRecord {
    using V4 = typename Vec4<T>::type;
    static constexpr size_t kPackets = sizeof(T) == 4 ? 5 : 4;
    static __host__ __device__ size_t bytes(size_t n) { return n * (kPackets * sizeof(V4) + 24) + ((4 * n + 7) & ~size_t(7)) + ((n + 7) & ~size_t(7)); }
    static __host__ __device__ V4* pk0(void* b, size_t) { return (V4*)b; }
    static __host__ __device__ V4* pk1(void* b, size_t n) { return (V4*)b + n; }
    static __host__ __device__ V4* acc(void* b, size_t n) { return (V4*)b + 2 * n; }
    static __host__ __device__ V4* ghost(void* b, size_t n) { return (V4*)b + 3 * n; }
    static __host__ __device__ V4* comp(void* b, size_t n) { return (V4*)b + 4 * n; }                  // (fp32 records)
    static __host__ __device__ long long* id(void* b, size_t n) { return (long long*)((V4*)b + kPackets * n); }
    static __host__ __device__ unsigned long long* grp(void* b, size_t n) { return (unsigned long long*)id(b, n) + n; }
    static __host__ __device__ unsigned long long* tag(void* b, size_t n) { return grp(b, n) + n; }
    static __host__ __device__ int* prow(void* b, size_t n) { return (int*)(tag(b, n) + n); }
    static __host__ __device__ uint8_t* type(void* b, size_t n) { return (uint8_t*)prow(b, n) + ((4 * n + 7) & ~size_t(7)); }
};

template <class T>
__global__ void __launch_bounds__(256) k_dd_gather(Half<const typename Vec4<T>::type> pk0, Half<const typename Vec4<T>::type> pk1,
                                                   const typename Vec4<T>::type* acc, const typename Vec4<T>::type* ghost,
                                                   const typename Vec4<T>::type* comp, const long long* id, const unsigned long long* grp,
                                                   const unsigned long long* tag, const int* prow, const uint8_t* type,
                                                   const int* idx, int n, void* buf) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = idx[k];
    DdRecord<T>::pk0(buf, n)[k] = pk0[i];
    DdRecord<T>::pk1(buf, n)[k] = pk1[i];
    DdRecord<T>::acc(buf, n)[k] = acc[i];
    DdRecord<T>::ghost(buf, n)[k] = ghost[i];
    if constexpr (sizeof(T) == 4) {
        typename Vec4<T>::type c; c.x = c.y = c.z = c.w = T(0);
        if (comp) c = comp[i];
        DdRecord<T>::comp(buf, n)[k] = c;
    }
    DdRecord<T>::id(buf, n)[k] = id[i];
    DdRecord<T>::grp(buf, n)[k] = grp[i];
    DdRecord<T>::tag(buf, n)[k] = tag[i];
    DdRecord<T>::prow(buf, n)[k] = prow[i];
    DdRecord<T>::type(buf, n)[k] = type[i] & kTypeMask;
}

template <class T>
__global__ void __launch_bounds__(256) k_dd_append(Half<typename Vec4<T>::type> pk0, Half<typename Vec4<T>::type> pk1,
                                                   typename Vec4<T>::type* acc, typename Vec4<T>::type* ghost, typename Vec4<T>::type* comp, long long* id,
                                                   unsigned long long* grp, unsigned long long* tag, int* prow, uint8_t* type, int at, int n,
                                                   void* buf, uint8_t flag) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    pk0[at + k] = DdRecord<T>::pk0(buf, n)[k];
    pk1[at + k] = DdRecord<T>::pk1(buf, n)[k];
    acc[at + k] = DdRecord<T>::acc(buf, n)[k];
    ghost[at + k] = DdRecord<T>::ghost(buf, n)[k];
    if constexpr (sizeof(T) == 4) { if (comp) comp[at + k] = DdRecord<T>::comp(buf, n)[k]; }
    id[at + k] = DdRecord<T>::id(buf, n)[k];
    grp[at + k] = DdRecord<T>::grp(buf, n)[k];
    tag[at + k] = DdRecord<T>::tag(buf, n)[k];
    prow[at + k] = DdRecord<T>::prow(buf, n)[k];
    type[at + k] = DdRecord<T>::type(buf, n)[k] | flag;
}

__global__ void __launch_bounds__(256) k_dd_kill(uint8_t* type, const int* idx, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) type[idx[k]] = 0;
}
__global__ void __launch_bounds__(256) k_dd_kill_ghosts(uint8_t* type, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N && (type[i] & kGhostMask)) type[i] = 0;
}

}  // namespace sphmi
