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

namespace sphmi {

template <class T> struct Vec4;
template <> struct Vec4<float>  { using type = float4;  };
template <> struct Vec4<double> { using type = double4; };

constexpr int kWave = 64;
constexpr int kSlots = 32;        // mask slots (64 candidates each) buffered in LDS per wave
constexpr int kChunkGroup = 4;    // candidate chunks held in registers at once

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

// v_writelane_b32 pair: lo[lane] = vlo, hi[lane] = vhi (vlo, vhi, lane wave-uniform).  clang exposes
// no builtin for it.  On gfx9-class encodings a VOP3 may read one SGPR only, so the lane select goes
// through M0; M0 is compiler-reserved, hence saved and restored inside the statement.
__device__ __forceinline__ void writelane2(int& lo, int& hi, int vlo, int vhi, int lane) {
    int keep;
    asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
        "v_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
        : "+v"(lo), "+v"(hi), "=&s"(keep)
        : "s"(vlo), "s"(vhi), "s"(lane));
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
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, double v) {
    atomicMax(p, (unsigned long long)__double_as_longlong(v));
}

// ------------------------------------------------------------------------------------------
// The neighbour + force kernel.
// ------------------------------------------------------------------------------------------
template <class T, int D, int PASS>
__global__ void __launch_bounds__(kWave)
k_neighbor_force(const ForceParams<T> P) {
    using V4 = typename Vec4<T>::type;
    constexpr int NSEG = (D == 3) ? 9 : 3;
    __shared__ unsigned long long s_mask[kSlots * kWave];
    __shared__ int s_base[kSlots];

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

    const int key_a = P.key[ac];
    const int cs_a = P.cstart[key_a], ce_a = P.cstart[key_a + 1];
    const int last_lane = min(kWave - 1, P.N - 1 - t0);

    T drho = 0, ax = 0, ay = 0, az = 0;
    int nslots = 0;

    // ---- phase 2: walk the buffered accept masks -------------------------------------------
    auto drain = [&]() {
        __syncthreads();
        int s = 0, base = 0;
        unsigned long long m = 0;
        while (true) {
            while (m == 0 && s < nslots) {
                m = s_mask[s * kWave + lane];
                base = s_base[s];
                ++s;
            }
            if (m == 0) break;
            const int j = base + __builtin_ctzll(m);
            m &= m - 1;
            const V4 n0 = P.src0[j];
            const V4 n1 = P.src1[j];
            const T dx = xa - n0.x, dy = ya - n0.y, dz = za - n0.z;
            const T r2 = dx * dx + dy * dy + dz * dz;
            T rho_b, rhon_b, P_b, s_b;
            if constexpr (PASS == PASS_CORRECTOR) {
                rho_b = n0.w; rhon_b = absT(n1.w); s_b = n1.w;
                P_b = eos7<T>(rho_b, P.rho0, P.inv_rho0, P.Cbe);
            } else {
                rho_b = absT(n0.w); rhon_b = rho_b; s_b = n0.w; P_b = n1.w;
            }
            // ∇W factor, src/SPHKernels.jl:80-87 with q = clamp(r/h, 0, 2) (src/SPHCellList.jl:280)
            const T r = fast_sqrt(r2);
            T qq = r * P.h_inv;
            qq = qq > T(2) ? T(2) : qq;
            const T tq = qq - T(2);
            const T fac = P.Cgw * (tq * tq * tq);
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
        }
        nslots = 0;
        __syncthreads();
    };

    // ---- phase 1: accept masks, one cell row (3 x-adjacent cells per target) at a time ------
#pragma unroll 1
    for (int seg = 0; seg < NSEG; ++seg) {
        const int off = (D == 3) ? ((seg % 3) - 1) * P.nxp + ((seg / 3) - 1) * P.nxyp
                                 : (seg - 1) * P.nxp;
        const int lo_l = valid ? P.cstart[key_a + off - 1] : 0x7fffffff;
        const int hi_l = valid ? P.cstart[key_a + off + 2] : 0;
        // keys are sorted, cstart is monotone: the union of the lanes' ranges is [lo(first), hi(last))
        const int LO = rl_i(lo_l, 0);
        const int HI = rl_i(hi_l, last_lane);
#pragma unroll 1
        for (int gbase = LO; gbase < HI; gbase += kChunkGroup * kWave) {
            const int rem = HI - gbase;
            const int nch = rem >= kChunkGroup * kWave ? kChunkGroup : (rem + kWave - 1) / kWave;
            if (nslots + nch > kSlots) drain();
            T cx[kChunkGroup], cy[kChunkGroup], cz[kChunkGroup];
            int mlo[kChunkGroup], mhi[kChunkGroup];
#pragma unroll
            for (int k = 0; k < kChunkGroup; ++k) {
                const int c = gbase + k * kWave + lane;
                mlo[k] = 0; mhi[k] = 0;
                if (k < nch && c < HI) {
                    const V4 cpk = P.src0[c];
                    cx[k] = cpk.x; cy[k] = cpk.y; cz[k] = cpk.z;
                } else {
                    cx[k] = T(1e30); cy[k] = T(1e30); cz[k] = T(1e30);
                }
            }
#pragma unroll 4
            for (int t = 0; t <= last_lane; ++t) {
                const T tx = rl(xa, t), ty = rl(ya, t), tz = rl(za, t);
                const int tlo = rl_i(lo_l, t);
                const unsigned tw = (unsigned)(rl_i(hi_l, t) - tlo);
#pragma unroll
                for (int k = 0; k < kChunkGroup; ++k) {
                    if (k < nch) {
                        const T ex = cx[k] - tx, ey = cy[k] - ty, ez = cz[k] - tz;
                        const T r2 = ex * ex + ey * ey + ez * ez;
                        const int c = gbase + k * kWave + lane;
                        const bool in = (r2 <= P.H2) && ((unsigned)(c - tlo) < tw);
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
                        writelane2(mlo[k], mhi[k], (int)(unsigned)(bal & 0xffffffffull), (int)(unsigned)(bal >> 32), t);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kChunkGroup; ++k) {
                if (k < nch) {
                    s_mask[(nslots + k) * kWave + lane] =
                        ((unsigned long long)(unsigned)mhi[k] << 32) | (unsigned)mlo[k];
                    if (lane == 0) s_base[nslots + k] = gbase + k * kWave;
                }
            }
            nslots += nch;
        }
    }
    drain();

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
