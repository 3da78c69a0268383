// sphmi_kernels.h — gfx950 (CDNA4, wave64) device kernels of the SPH neighbour + force engine.
//
// Data layout in HBM (SoA of 16-/32-byte packets, cell-sorted, x-fastest cell order):
//   pk0[i] = { x, y, z, ρ·s }   s = +1 for Fluid (MotionLimiter = 1), −1 otherwise
//   pk1[i] = { vx, vy, vz, P }  (state set "A"/"B")   or   { v⁺, ρⁿ·s } (half-step set "H")
// 2-D runs keep z = vz = 0 and use component D−1 as the gravity / hydrostatic axis.
//
// The neighbour kernel replaces NeighborLoop! + ComputeInteractions! + ReductionStep! + HalfTimeStep /
// FullTimeStep + LimitDensityAtBoundary! + DensityEpsi! + Pressure! + the Δt / Δx reductions of the
// reference (src/SPHCellList.jl:168-217, 268-317, 367-381, 624-652, 706-724;
// src/SimulationEquations.jl:9-42; src/TimeStepping.jl:24-46) with ONE launch per pass.
//
// Mapping (one wave = 64 consecutive sorted target particles, one wave per workgroup):
//   phase 1  "who is within H": for each of the 3^(D-1) cell rows around the targets the three
//            x-adjacent cells are one contiguous particle range (x is the fastest sort axis), so the
//            candidates of a row are loaded coalesced, ONE CANDIDATE PER LANE; the 64 targets are
//            broadcast one after the other through SGPRs (v_readlane) and every distance test is a
//            v_cmp whose 64-bit result is already the compacted accept mask of that target.  The mask
//            goes to the target's lane (v_writelane) and then to LDS.
//   phase 2  "pair physics": every lane walks the set bits of its own masks, gathers the accepted
//            neighbour packets and accumulates dρ/dt and acceleration — no divergence on the accept
//            branch, no atomics, each output written once.
//   epilogue predictor or corrector fused in; wave-level max-reductions for Δt / Δx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace sphmi {

template <class T> struct Vec4;
template <> struct Vec4<float>  { using type = float4;  };
template <> struct Vec4<double> { using type = double4; };

constexpr int kWave = 64;
#ifndef SPHMI_KSLOTS
#define SPHMI_KSLOTS 32
#endif
#ifndef SPHMI_ABL_NO_TLOOP
#define SPHMI_ABL_NO_TLOOP 0
#endif
#ifndef SPHMI_ABL_RL
#define SPHMI_ABL_RL 0
#endif
#ifndef SPHMI_ABL_WL
#define SPHMI_ABL_WL 0
#endif
#ifndef SPHMI_ABL_NO_CONSUME
#define SPHMI_ABL_NO_CONSUME 0
#endif
#ifndef SPHMI_PIPE
#define SPHMI_PIPE 0
#endif
#ifndef SPHMI_ABL_NO_P2
#define SPHMI_ABL_NO_P2 0
#endif
#ifndef SPHMI_RING_ROWS
#define SPHMI_RING_ROWS 4
#endif
#ifndef SPHMI_CHUNKS
#define SPHMI_CHUNKS 2
#endif
constexpr int kChunkGroup = SPHMI_CHUNKS;   // candidate chunks (64 each) per cell row held in registers / LDS slots

static_assert(kChunkGroup == 1 || kChunkGroup == 2 || kChunkGroup == 4, "SPHMI_CHUNKS must be 1, 2 or 4");
constexpr int kLogChunks = kChunkGroup == 4 ? 2 : (kChunkGroup == 2 ? 1 : 0);

enum { PASS_FORCES_ONLY = 0, PASS_PREDICTOR = 1, PASS_CORRECTOR = 2 };

template <class T>
struct ForceParams {
    using V4 = typename Vec4<T>::type;
    const V4* src0;      // neighbour stream packet 0 (A for pass 1, H for pass 2)
    const V4* src1;      // neighbour stream packet 1
    const V4* a0;        // state A (corrector epilogue)
    const V4* a1;
    V4* out0;            // H (predictor) or B (corrector)
    V4* out1;
    V4* accbuf;          // { a, dρ/dt }
    const int* key;      // padded linear cell id of every sorted particle
    const int* cstart;   // exclusive scan of cell counts, ncell+1 entries
    const uint8_t* type;
    unsigned long long* red;   // [0] max |x⁺−x|², [1] max visc, [2] max |a|² (bit patterns), [3] bad-ρ flag
    int N, nxp, nxyp, nblocks;
    int visc, ddt;
    T dt, dt2;
    T H2, h, h_inv, Cgw, m0, Kddt, linfac, eta2, Kv2, rho0, inv_rho0, g, Cbe;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float  fast_rcp(float x)  { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float  fast_sqrt(float x)  { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }

__device__ __forceinline__ float  absT(float x)  { return __builtin_fabsf(x); }
__device__ __forceinline__ double absT(double x) { return __builtin_fabs(x); }

// v_writelane_b32: dst[lane] = value (value, lane wave-uniform).  clang exposes no builtin for it.
// On gfx9-class encodings a VOP3 may read one SGPR only, so the lane select goes through M0; M0 is
// compiler-reserved, hence saved and restored inside the statement (one statement per target lane
// covers the masks of all NCH candidate chunks).
template <int NCH>
__device__ __forceinline__ void writelanes(int (&lo)[NCH], int (&hi)[NCH], const int (&vlo)[NCH],
                                           const int (&vhi)[NCH], int lane) {
    int keep;
    if constexpr (NCH == 1) {
        asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
            : "+v"(lo[0]), "+v"(hi[0]), "=&s"(keep)
            : "s"(vlo[0]), "s"(vhi[0]), "s"(lane));
    } else if constexpr (NCH == 2) {
        asm("s_mov_b32 %4, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\t"
            "v_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0\n\ts_mov_b32 m0, %4"
            : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "=&s"(keep)
            : "s"(vlo[0]), "s"(vhi[0]), "s"(vlo[1]), "s"(vhi[1]), "s"(lane));
    } else if constexpr (NCH == 3) {
        asm("s_mov_b32 %6, m0\n\ts_mov_b32 m0, %13\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %7, m0\n\tv_writelane_b32 %1, %8, m0\n\t"
            "v_writelane_b32 %2, %9, m0\n\tv_writelane_b32 %3, %10, m0\n\t"
            "v_writelane_b32 %4, %11, m0\n\tv_writelane_b32 %5, %12, m0\n\ts_mov_b32 m0, %6"
            : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "=&s"(keep)
            : "s"(vlo[0]), "s"(vhi[0]), "s"(vlo[1]), "s"(vhi[1]), "s"(vlo[2]), "s"(vhi[2]), "s"(lane));
    } else {
        static_assert(NCH == 4, "writelanes: 1..4 chunks");
        asm("s_mov_b32 %8, m0\n\ts_mov_b32 m0, %17\n\ts_nop 0\n\t"
            "v_writelane_b32 %0, %9, m0\n\tv_writelane_b32 %1, %10, m0\n\t"
            "v_writelane_b32 %2, %11, m0\n\tv_writelane_b32 %3, %12, m0\n\t"
            "v_writelane_b32 %4, %13, m0\n\tv_writelane_b32 %5, %14, m0\n\t"
            "v_writelane_b32 %6, %15, m0\n\tv_writelane_b32 %7, %16, m0\n\ts_mov_b32 m0, %8"
            : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]),
              "+v"(hi[3]), "=&s"(keep)
            : "s"(vlo[0]), "s"(vhi[0]), "s"(vlo[1]), "s"(vhi[1]), "s"(vlo[2]), "s"(vhi[2]), "s"(vlo[3]),
              "s"(vhi[3]), "s"(lane));
    }
}

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
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, float v) {
    unsigned int* q = reinterpret_cast<unsigned int*>(p);
    const unsigned int b = __float_as_uint(v);
    if (b > __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(q, b);
}
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
}

// ------------------------------------------------------------------------------------------
// The neighbour + force kernel.
// ------------------------------------------------------------------------------------------
template <class T, int D, int PASS>
__global__ void __launch_bounds__(kWave)
k_neighbor_force(const ForceParams<T> P) {
    using V4 = typename Vec4<T>::type;
    constexpr int NSEG = (D == 3) ? 9 : 3;
    constexpr int RB = SPHMI_RING_ROWS;                    // cell rows buffered in the LDS ring
    static_assert((RB & (RB - 1)) == 0 && (kChunkGroup & (kChunkGroup - 1)) == 0, "ring geometry: powers of two");
    constexpr int NSLOT = RB * kChunkGroup;
    __shared__ unsigned long long s_mask[NSLOT * kWave];   // [row % RB][chunk][lane] accept masks
    __shared__ int s_rowbase[RB * kWave];                  // [row % RB][lane] candidate index of bit 0, chunk 0

    const int lane = threadIdx.x;
    // XCD-aware block order: the dispatcher places block b on XCD b % 8; give every XCD a contiguous
    // run of target tiles so neighbouring tiles (which share their source rows) share one L2.
    int b = blockIdx.x;
    {
        const int nb = P.nblocks, per = nb >> 3;
        if (per > 0 && b < per * 8) b = (b & 7) * per + (b >> 3);
    }
    const int t0 = b * kWave;
    const int a = t0 + lane;
    const bool valid = a < P.N;
    const int ac = valid ? a : P.N - 1;

    // target data
    const V4 q0 = P.src0[ac];
    const V4 q1 = P.src1[ac];
    const T xa = q0.x, ya = q0.y, za = q0.z;
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
    const bool fluid_a = s_a > T(0);
    const T inv_rho_a = fast_rcp(rho_a);
    const T inv_rhon_a = (PASS == PASS_CORRECTOR) ? fast_rcp(rhon_a) : inv_rho_a;
    const T rm_a = rho_a * P.m0;

    const int key_a = valid ? P.key[ac] : -1;
    const int cs_a = P.cstart[valid ? key_a : 0], ce_a = P.cstart[valid ? key_a + 1 : 0];
    const int last_lane = min(kWave - 1, P.N - 1 - t0);

    // Lanes are sorted by cell: a "run" = the lanes of one cell.  All targets of a run share the same
    // candidate range per cell row, so the distance tests need no per-target range check, and a local
    // origin (the run's first particle) makes the expanded form |c|² − 2c·t + |t|² safe in fp32.
    const int key_prev = __shfl_up(key_a, 1, kWave);
    const unsigned long long heads = __builtin_amdgcn_ballot_w64(valid && (lane == 0 || key_a != key_prev));
    const int r0_l = 63 - __builtin_clzll((heads & (~0ull >> (63 - lane))) | 1ull);
    const T oxl = __shfl(xa, r0_l, kWave), oyl = __shfl(ya, r0_l, kWave), ozl = __shfl(za, r0_l, kWave);
    const T txl = xa - oxl, tyl = ya - oyl, tzl = za - ozl;
    // accept  |c − t|² ≤ H²(1+ε)  ⇔  |c|² − 2c·t ≤ H²(1+ε) − |t|² ; the exact test is redone in phase 2
    const T m2x = T(-2) * txl, m2y = T(-2) * tyl, m2z = T(-2) * tzl;
    const T thr = P.H2 * T(1.0 + 1.0 / 1024.0) - (txl * txl + tyl * tyl + tzl * tzl);

    T drho = 0, ax = 0, ay = 0, az = 0;

    // ---- pair physics for one accepted neighbour j ------------------------------------------
    auto pair = [&](const int j, const V4& n0, const V4& n1) {
        const T dx = xa - n0.x, dy = ya - n0.y, dz = za - n0.z;
        const T r2 = dx * dx + dy * dy + dz * dz;
        T rho_b, rhon_b, P_b, s_b;
        if constexpr (PASS == PASS_CORRECTOR) {
            rho_b = n0.w; rhon_b = absT(n1.w); s_b = n1.w;
            P_b = eos7<T>(rho_b, P.rho0, P.inv_rho0, P.Cbe);
        } else {
            rho_b = absT(n0.w); rhon_b = rho_b; s_b = n0.w; P_b = n1.w;
        }
        // ∇W factor, src/SPHKernels.jl:80-87 with q = clamp(r/h, 0, 2) (src/SPHCellList.jl:280);
        // the phase-1 mask is slightly generous, the exact r² ≤ H² test of :275 is applied here
        const T r = fast_sqrt(r2);
        T qq = r * P.h_inv;
        qq = qq > T(2) ? T(2) : qq;
        const T tq = qq - T(2);
        T fac = P.Cgw * (tq * tq * tq);
        fac = r2 <= P.H2 ? fac : T(0);
        const T dvx = q1.x - n1.x, dvy = q1.y - n1.y, dvz = q1.z - n1.z;
        const T vdx = dvx * dx + dvy * dy + dvz * dz;          // vᵢⱼ·xᵢⱼ
        const T inv_rho_b = fast_rcp(rho_b);
        // continuity, src/SPHCellList.jl:289-291 (both orientations give the same target term)
        drho += rm_a * inv_rho_b * (fac * vdx);
        const T inv_r2e = fast_rcp(r2 + P.eta2);
        if (P.ddt) {
            // LinearDensityDiffusion, src/SPHDensityDiffusionModels.jl:116-133; orientation rule
            // of SURVEY §8(a)-Q4: target plays "i" iff j sorts before its cell, or after it inside it
            const T dlast = (D == 3) ? dz : dy;
            const T drn = (rhon_b - rhon_a) - P.linfac * dlast;
            const T psigw = T(-2) * drn * fac * r2 * inv_r2e;
            const bool a_is_i = (j < cs_a) || (j > a && j < ce_a);
            T inv_sel;
            if constexpr (PASS == PASS_CORRECTOR) inv_sel = a_is_i ? fast_rcp(rhon_b) : inv_rhon_a;
            else inv_sel = a_is_i ? inv_rho_b : inv_rho_a;
            const T Dv = P.Kddt * inv_sel * psigw;
            drho += (fluid_a && s_b > T(0)) ? Dv : T(0);
        }
        // pressure, src/SPHCellList.jl:301-303 (tensile term is 0 for Wendland)
        T coef = -P.m0 * ((P_a + P_b) * inv_rho_a * inv_rho_b);
        if (P.visc) {
            // ArtificialViscosity, src/SPHViscosityModels.jl:56-74 (ρ̄ from SimParticles.Density)
            const T vneg = vdx < T(0) ? vdx : T(0);
            coef += P.Kv2 * vneg * inv_r2e * fast_rcp(rhon_a + rhon_b);
        }
        coef *= fac;
        ax += coef * dx; ay += coef * dy; az += coef * dz;
    };

    // ---- phase 2: every lane walks the set bits of its own accept masks -----------------------
    // The masks of the last RB cell rows live in an LDS ring ([row % RB][chunk][lane]).  Lanes consume
    // at their own pace: a lane with few neighbours in the old rows runs ahead into the newer ones
    // instead of idling, and a row slot is only recycled once EVERY lane is through with it.
    int cs = 0;                      // next slot (absolute: row * K + chunk) this lane will fetch
    int cbase = 0;                   // candidate index of bit 0 of the current mask
    unsigned long long cm = 0;       // unconsumed bits of the current mask
#if SPHMI_PIPE
    // one gathered pair in flight per lane: { pj, p0, p1 } was taken from slot pslot and is evaluated one
    // iteration later, so the gather latency overlaps the previous pair's arithmetic
    bool phave = false;
    int pj = 0, pslot = 0;
    V4 p0{}, p1{};
#endif
    // consume until every lane has finished all slots below `upto` (slots < produced are readable)
    auto consume = [&](const int upto, const int produced) {
#if SPHMI_ABL_NO_CONSUME
        ax += T(s_mask[lane] & 1);
        return;
#endif
        __syncthreads();
        while (true) {
            const bool empty = cm == 0;
            // a lane still owes old work if it has not fetched all slots < upto, or is inside one of them
            bool owes = (cs < upto) | (!empty & (cs <= upto));
#if SPHMI_PIPE
            owes |= phave & (pslot < upto);
#endif
            if (!__builtin_amdgcn_ballot_w64(owes)) break;
            if (empty & (cs < produced)) {
                cm = s_mask[(cs & (NSLOT - 1)) * kWave + lane];
                cbase = s_rowbase[((cs >> kLogChunks) & (RB - 1)) * kWave + lane] + ((cs & (kChunkGroup - 1)) << 6);
                ++cs;
            }
#if SPHMI_PIPE
            const bool chave = phave;
            const int cj = pj;
            const V4 c0 = p0, c1 = p1;
            phave = cm != 0;
            if (phave) {
                pj = cbase + __builtin_ctzll(cm);
                cm &= cm - 1;
                pslot = cs - 1;
                p0 = P.src0[pj];
                p1 = P.src1[pj];
            }
            if (chave) pair(cj, c0, c1);
#else
            if (cm != 0) {
                const int j = cbase + __builtin_ctzll(cm);
                cm &= cm - 1;
#if SPHMI_ABL_NO_P2
                ax += T(j);
#else
                const V4 n0 = P.src0[j];
                const V4 n1 = P.src1[j];
                pair(j, n0, n1);
#endif
            }
#endif
        }
        __syncthreads();
    };

    // ---- phase 1: accept masks --------------------------------------------------------------
    // For one run and one cell row: NCH candidate chunks (one candidate per lane each, coalesced
    // loads) live in registers while the run's targets are broadcast through SGPRs; the 64-bit compare
    // result of target t IS its accept mask and is lane-transposed into lane t's registers.
    auto scan = [&](auto nch_tag, const int gb, const int hi, const int r0, const int r1,
                    int (&mlo)[kChunkGroup], int (&mhi)[kChunkGroup]) {
        constexpr int NCH = decltype(nch_tag)::value;
        const T ox = rl(xa, r0), oy = rl(ya, r0), oz = rl(za, r0);
        T cx[NCH], cy[NCH], cz[NCH], cc[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = gb + k * kWave + lane;
            if (c < hi) {
                const V4 cpk = P.src0[c];
                cx[k] = cpk.x - ox; cy[k] = cpk.y - oy; cz[k] = cpk.z - oz;
                cc[k] = cx[k] * cx[k] + cy[k] * cy[k] + cz[k] * cz[k];
            } else {
                cx[k] = T(0); cy[k] = T(0); cz[k] = T(0); cc[k] = T(1e30);
            }
        }
        int wlo[NCH], whi[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) { wlo[k] = mlo[k]; whi[k] = mhi[k]; }
#if SPHMI_ABL_NO_TLOOP
        for (int t = r0; t < r0; ++t) {
#else
#pragma unroll 2
        for (int t = r0; t < r1; ++t) {
#endif
#if SPHMI_ABL_RL
            const T sx = rl(m2x, r0) + T(t), sy = rl(m2y, r0), sz = rl(m2z, r0), st = rl(thr, r0);
#else
            const T sx = rl(m2x, t), sy = rl(m2y, t), sz = rl(m2z, t), st = rl(thr, t);
#endif
            int blo[NCH], bhi[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                T d = cx[k] * sx + cc[k];
                d = cy[k] * sy + d;
                d = cz[k] * sz + d;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(d <= st);
                blo[k] = (int)(unsigned)(bal & 0xffffffffull);
                bhi[k] = (int)(unsigned)(bal >> 32);
            }
#if SPHMI_ABL_WL == 2
#pragma unroll
            for (int k = 0; k < NCH; ++k) { wlo[k] |= (lane == 0) ? blo[k] : 0; whi[k] ^= (lane == 1) ? bhi[k] : 0; }
#elif SPHMI_ABL_WL == 1
            writelanes<NCH>(wlo, whi, blo, bhi, 0);
#else
            writelanes<NCH>(wlo, whi, blo, bhi, t);
#endif
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) { mlo[k] = wlo[k]; mhi[k] = whi[k]; }
    };

#pragma unroll 1
    for (int g = 0;; ++g) {          // chunk-group passes: g > 0 only when a row range exceeds K·64 candidates
        bool more = false;
        cs = 0; cm = 0;   // (the pair pipeline is empty here: the previous pass was drained)
#pragma unroll 1
        for (int seg = 0; seg < NSEG; ++seg) {
            // recycle ring position seg % RB: every lane must be through row seg − RB
            if (seg >= RB) consume((seg - RB + 1) * kChunkGroup, seg * kChunkGroup);
            const int off = (D == 3) ? ((seg % 3) - 1) * P.nxp + ((seg / 3) - 1) * P.nxyp
                                     : (seg - 1) * P.nxp;
            const int lo_l = valid ? P.cstart[key_a + off - 1] : 0;
            const int hi_l = valid ? P.cstart[key_a + off + 2] : 0;
            int mlo[kChunkGroup], mhi[kChunkGroup];
#pragma unroll
            for (int k = 0; k < kChunkGroup; ++k) { mlo[k] = 0; mhi[k] = 0; }
#pragma unroll 1
            for (int r0 = 0; r0 <= last_lane;) {
                const unsigned long long rest = heads >> r0 >> 1;          // heads after r0
                const int r1 = rest ? r0 + 1 + __builtin_ctzll(rest) : last_lane + 1;
                const int lo = rl_i(lo_l, r0), hi = rl_i(hi_l, r0);
                const int gb = lo + g * kChunkGroup * kWave;
                const int rem = hi - gb;
                if (rem > 0) {
                    if (rem > kChunkGroup * kWave) more = true;
                    const int nch = rem >= kChunkGroup * kWave ? kChunkGroup : (rem + kWave - 1) / kWave;
                    if (nch == 1) scan(std::integral_constant<int, 1>{}, gb, hi, r0, r1, mlo, mhi);
                    if constexpr (kChunkGroup >= 2) if (nch == 2) scan(std::integral_constant<int, 2>{}, gb, hi, r0, r1, mlo, mhi);
                    if constexpr (kChunkGroup >= 3) if (nch == 3) scan(std::integral_constant<int, 3>{}, gb, hi, r0, r1, mlo, mhi);
                    if constexpr (kChunkGroup >= 4) if (nch == 4) scan(std::integral_constant<int, 4>{}, gb, hi, r0, r1, mlo, mhi);
                }
                r0 = r1;
            }
            s_rowbase[(seg % RB) * kWave + lane] = lo_l + g * kChunkGroup * kWave;
#pragma unroll
            for (int k = 0; k < kChunkGroup; ++k)
                s_mask[((seg % RB) * kChunkGroup + k) * kWave + lane] =
                    ((unsigned long long)(unsigned)mhi[k] << 32) | (unsigned)mlo[k];
        }
        consume(NSEG * kChunkGroup, NSEG * kChunkGroup);
        if (!more) break;
    }

    // ---- epilogue ---------------------------------------------------------------------------
    const uint8_t ty_a = P.type[ac];
    const T gf = ty_a == 1 ? T(-1) : (ty_a == 3 ? T(1) : T(0));     // src/PreProcess.jl:78-87
    const T ml = fluid_a ? T(1) : T(0);
    if constexpr (PASS == PASS_FORCES_ONLY) {
        V4 o; o.x = ax; o.y = ay; o.z = az; o.w = drho;
        if (valid) P.accbuf[a] = o;
    } else if constexpr (PASS == PASS_PREDICTOR) {
        // HalfTimeStep (src/SPHCellList.jl:624-638) + LimitDensityAtBoundary! (SimulationEquations.jl:36-42)
        if constexpr (D == 3) az += P.g * gf; else ay += P.g * gf;
        V4 o0, o1;
        o0.x = xa + q1.x * P.dt2 * ml; o0.y = ya + q1.y * P.dt2 * ml; o0.z = za + q1.z * P.dt2 * ml;
        o1.x = q1.x + ax * P.dt2 * ml; o1.y = q1.y + ay * P.dt2 * ml; o1.z = q1.z + az * P.dt2 * ml;
        T rho_h = rho_a + drho * P.dt2;
        if (!fluid_a && rho_h < P.rho0) rho_h = P.rho0;
        o0.w = rho_h;
        o1.w = s_a;                         // ρⁿ·s travels with the half-step stream
        if (valid) { P.out0[a] = o0; P.out1[a] = o1; }
    } else {
        // LimitDensityAtBoundary!(Density) → DensityEpsi! → FullTimeStep
        // (src/SPHCellList.jl:794-798, 640-652; src/SimulationEquations.jl:28-33)
        const V4 s0 = P.a0[ac];
        const V4 s1 = P.a1[ac];
        T rho_n = absT(s0.w);
        if (!fluid_a && rho_n < P.rho0) rho_n = P.rho0;
        const T epsi = -(drho / rho_a) * P.dt;
        const T rho_new = rho_n * ((T(2) - epsi) / (T(2) + epsi));
        if constexpr (D == 3) az += P.g * gf; else ay += P.g * gf;
        const T adx = ax * P.dt * ml, ady = ay * P.dt * ml, adz = az * P.dt * ml;
        V4 o0, o1, oa;
        o1.x = s1.x + adx; o1.y = s1.y + ady; o1.z = s1.z + adz;
        o0.x = s0.x + (((o1.x + (o1.x - adx)) / T(2)) * P.dt) * ml;
        o0.y = s0.y + (((o1.y + (o1.y - ady)) / T(2)) * P.dt) * ml;
        o0.z = s0.z + (((o1.z + (o1.z - adz)) / T(2)) * P.dt) * ml;
        o0.w = fluid_a ? rho_new : -rho_new;
        o1.w = eos7<T>(rho_new, P.rho0, P.inv_rho0, P.Cbe);
        oa.x = ax; oa.y = ay; oa.z = az; oa.w = drho;
        if (valid) {
            P.out0[a] = o0; P.out1[a] = o1; P.accbuf[a] = oa;
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
        if (!valid) { disp2 = T(0); vis = T(0); a2 = T(0); }      // tail lanes of the last tile
        disp2 = wave_max(disp2); vis = wave_max(vis); a2 = wave_max(a2);
        if (lane == 0) {
            atomic_max_bits(&P.red[0], disp2);
            atomic_max_bits(&P.red[1], vis);
            atomic_max_bits(&P.red[2], a2);
        }
    }
}

}  // namespace sphmi
