/*
 * sph_oracle.c — CPU restatement (plain C, fp64) of the SPHExample hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sphexample_amd/ may import, link or execute this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the timed CPU baseline ("port"), never as the product path.
 *
 * PARITY STATUS: "parity unpinned" for the pair path.  The reference (Julia >= 1.11) cannot be
 * built or run in this image (no julia binary, no package depot, no network), so there is no
 * oracle/_ref.  Upstream's own tests (test/runtests.jl:6-16 and :18-75) pin only the time-step
 * criterion and an isolated particle; this restatement is checked against those two tests, the
 * closed-form two-particle values of SURVEY.md appendix A and an independent O(N^2) numpy
 * enumeration (tests/test_oracle.py).  The pin by the real reference is staged, not done: tools/dump_fixture.jl
 * (Julia >= 1.11) writes its state after one output interval on five layouts to tests/golden/reference/ and
 * tests/test_reference_fixtures.py checks THIS file against them; until someone runs it the status stays unpinned.
 * All citations below are relative to /root/reference/.
 *
 * Each function names the reference lines it follows.  Known, documented deviation:
 *   Q5  the reference's end sentinel ParticleRanges[IndexCounter+1] = length(ParticleRanges) = N+2
 *       (src/SPHCellList.jl:160 with :840) makes the last sorted cell's range end at N+1, one past
 *       the arrays (read under @inbounds: undefined behaviour).  Here the sentinel is N+1 so the
 *       last range ends at N.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/sphmi.h"

#define MAXD 3

typedef struct orc_handle {
    sphmi_config cfg;
    int     D;
    int64_t N;
    int     nthreads;
    /* SimParticles fields (src/PreProcess.jl:114) */
    double  *pos, *vel, *acc, *rho, *press, *gf, *ml, *ghost;
    uint8_t *type;
    int64_t *id;
    int64_t *prow;   /* row of every particle at the last orc_download_permutation: one more column that sort! permutes */
    uint64_t *group;
    int64_t *cells;          /* N*D */
    /* support arrays (src/PreProcess.jl:121-158) — NOT permuted by the sort */
    double  *drhodt, *vel_np, *pos_np, *rho_np;
    /* SimThreadedArrays (src/PreProcess.jl:198-215) */
    double  **drhodt_thr, **acc_thr;
    /* cell list (src/SPHCellList.jl:840-843) */
    int64_t *ranges;         /* N+2, 1-based values, slot 0 == reference slot 1 (dummy) */
    int64_t *ucells;         /* (N+1)*D, slot 0 dummy (zero cell) */
    int64_t index_counter;   /* reference IndexCounter (1-based count incl. dummy) */
    /* scratch for the sort */
    int64_t *perm, *perm_tmp;
    void    *scratch;
    /* mdbc */
    double  *bgam, *Agam;
    /* PlanarShifting accumulators ∇Cᵢ (N*D) and ∇◌rᵢ (N) with their per-thread copies (src/PreProcess.jl:198-215) */
    double  *gradC, *divr, **gradC_thr, **divr_thr;
    /* StoreKernelOutput: Kernel (N), KernelGradient (N*D) and their per-thread copies */
    double  *kern, *kgrad, **kern_thr, **kgrad_thr;
    /* MotionDetails by GroupMarker (src/SimulationGeometry.jl:17-22) */
    int      n_motion;
    struct { uint64_t group; double vel, start, dur, dir[MAXD]; } motion[16];
    /* SimMetaData */
    int64_t iteration, n_rebuilds;
    double  total_time, last_dt, delta_x;
    int     uploaded;
    char    err[256];
} orc_handle;

/* ---------------------------------------------------------------------------------------------
 * SPHKernels.jl:75-78 (Wᵢⱼ) and :80-87 (∇Wᵢⱼ factor), Wendland C2.
 * ------------------------------------------------------------------------------------------- */
static inline double W_kernel(const sphmi_config *c, double q) {
    if (c->kernel == SPHMI_KERNEL_CUBIC_SPLINE) {                              /* :89-92 */
        double a = (0.0 <= q && q <= 1.0) ? (1.0 - 1.5 * q * q + 0.75 * q * q * q) : 0.0;
        double b = (1.0 < q && q <= 2.0) ? 0.25 * (2.0 - q) * (2.0 - q) * (2.0 - q) : 0.0;
        return c->alphaD * (a + b);
    }
    double t = 1.0 - q / 2.0;
    double t2 = t * t;
    return c->alphaD * (t2 * t2) * (2.0 * q + 1.0);
}
/* ∇Wᵢⱼ = factor(q, |xᵢⱼ|) · xᵢⱼ; the CubicSpline form divides by (|xᵢⱼ| + η²) as the reference does (:94-106) */
static inline double gradW_factor_r(const sphmi_config *c, double q, double r) {
    if (c->kernel == SPHMI_KERNEL_CUBIC_SPLINE) {
        double dWdq;
        if (0.0 <= q && q <= 1.0) dWdq = c->alphaD * (-3.0 * q + 2.25 * q * q);
        else if (1.0 < q && q <= 2.0) dWdq = c->alphaD * (-0.75) * (2.0 - q) * (2.0 - q);
        else dWdq = 0.0;
        return dWdq * c->h_inv / (r + c->eta2);
    }
    double t = q - 2.0;
    return c->alphaD * 5.0 * (t * t * t) / (8.0 * c->h * c->h);
}
static inline double W_wendland(const sphmi_config *c, double q) { return W_kernel(c, q); }
static inline double gradW_factor(const sphmi_config *c, double q) { return gradW_factor_r(c, q, q * c->h); }
/* tensile_correction, :108-126 — note Wᵢⱼ(instance, dx): the reference passes dx where q is expected */
static inline double tensile_correction(const sphmi_config *c, double Pi, double ri, double Pj, double rj, double q) {
    if (c->kernel != SPHMI_KERNEL_CUBIC_SPLINE) return 0.0;
    double w = W_kernel(c, q) / W_kernel(c, c->dx);
    double w2 = w * w;
    return c->cubic_eps * ((Pi / (ri * ri)) + (Pj / (rj * rj))) * (w2 * w2);
}

/* SimulationEquations.jl:9-11  EquationOfStateGamma7 */
static inline double eos_gamma7(double rho, double c0, double rho0) {
    double r = rho / rho0;
    double r2 = r * r, r4 = r2 * r2;
    return ((c0 * c0 * rho0) / 7.0) * (r4 * r2 * r - 1.0);
}

/* SPHCellList.jl:56-61  map_floor: round half away from zero via trunc(|x|*H⁻¹ + 0.5) */
static inline int64_t map_floor(double x, double inv_cutoff) {
    double s = (x > 0.0) - (x < 0.0);
    return (int64_t)s * (int64_t)trunc(fma(fabs(x), inv_cutoff, 0.5));
}

/* CartesianIndex isless: last axis most significant */
static inline int cell_cmp(const int64_t *a, const int64_t *b, int D) {
    for (int d = D - 1; d >= 0; --d) {
        if (a[d] < b[d]) return -1;
        if (a[d] > b[d]) return 1;
    }
    return 0;
}

/* SimulationEquations.jl:18-24 Pressure! */
static void pressure(orc_handle *o, double *press, const double *rho) {
    for (int64_t i = 0; i < o->N; ++i) press[i] = eos_gamma7(rho[i], o->cfg.c0, o->cfg.rho0);
}

/* SimulationEquations.jl:36-42 LimitDensityAtBoundary! */
static void limit_density_at_boundary(orc_handle *o, double *rho) {
    for (int64_t i = 0; i < o->N; ++i)
        if (rho[i] < o->cfg.rho0 && o->ml[i] == 0.0) rho[i] = o->cfg.rho0;
}

/* SimulationEquations.jl:28-33 DensityEpsi! */
static void density_epsi(orc_handle *o, double dt) {
    for (int64_t i = 0; i < o->N; ++i) {
        double epsi = -(o->drhodt[i] / o->rho_np[i]) * dt;
        o->rho[i] *= (2.0 - epsi) / (2.0 + epsi);
    }
}

/* TimeStepping.jl:24-46 Δt */
static double delta_t(orc_handle *o) {
    const int D = o->D;
    const double h = o->cfg.h, eta2 = o->cfg.eta2;
    double visc = -INFINITY;  /* maximum() of |.| values; N >= 1 */
    double dt1 = INFINITY;
    for (int64_t i = 0; i < o->N; ++i) {
        double vr = 0, rr = 0, aa = 0;
        for (int d = 0; d < D; ++d) {
            vr += o->vel[i * D + d] * o->pos[i * D + d];
            rr += o->pos[i * D + d] * o->pos[i * D + d];
            aa += o->acc[i * D + d] * o->acc[i * D + d];
        }
        double v = fabs(h * vr / (rr + eta2));
        if (v > visc) visc = v;
        double t = sqrt(h / sqrt(aa));
        if (t < dt1) dt1 = t;
    }
    double dt2 = h / (o->cfg.c0 + visc);
    return o->cfg.CFL * fmin(dt1, dt2);
}

/* SPHCellList.jl:706-724 update_delta_x! */
static double update_delta_x(orc_handle *o, double dx_acc) {
    const int D = o->D;
    double maxd = 0.0;
    for (int64_t i = 0; i < o->N; ++i) {
        double s = 0;
        for (int d = 0; d < D; ++d) {
            double t = o->pos_np[i * D + d] - o->pos[i * D + d];
            s += t * t;
        }
        double n = sqrt(s);
        if (n > maxd) maxd = n;
    }
    return dx_acc + 4.0 * maxd;
}

/* ---------------------------------------------------------------------------------------------
 * SPHCellList.jl:138-163 UpdateNeighbors!  (+ :118-123 ExtractCells!)
 * stable sort of every SimParticles field by Cells, then the range scan.
 * ------------------------------------------------------------------------------------------- */
static void permute_f64(orc_handle *o, double *a, int w) {
    double *t = (double *)o->scratch;
    for (int64_t i = 0; i < o->N; ++i)
        for (int d = 0; d < w; ++d) t[i * w + d] = a[o->perm[i] * w + d];
    memcpy(a, t, sizeof(double) * (size_t)o->N * w);
}
static void permute_i64(orc_handle *o, int64_t *a, int w) {
    int64_t *t = (int64_t *)o->scratch;
    for (int64_t i = 0; i < o->N; ++i)
        for (int d = 0; d < w; ++d) t[i * w + d] = a[o->perm[i] * w + d];
    memcpy(a, t, sizeof(int64_t) * (size_t)o->N * w);
}
static void permute_u8(orc_handle *o, uint8_t *a) {
    uint8_t *t = (uint8_t *)o->scratch;
    for (int64_t i = 0; i < o->N; ++i) t[i] = a[o->perm[i]];
    memcpy(a, t, (size_t)o->N);
}

static void merge_sort_perm(orc_handle *o) {
    /* bottom-up stable merge sort of perm[] by cells[perm] */
    const int D = o->D;
    const int64_t N = o->N;
    int64_t *src = o->perm, *dst = o->perm_tmp;
    for (int64_t w = 1; w < N; w *= 2) {
        for (int64_t lo = 0; lo < N; lo += 2 * w) {
            int64_t mid = lo + w < N ? lo + w : N;
            int64_t hi = lo + 2 * w < N ? lo + 2 * w : N;
            int64_t a = lo, b = mid, k = lo;
            while (a < mid && b < hi) {
                if (cell_cmp(&o->cells[src[b] * D], &o->cells[src[a] * D], D) < 0) dst[k++] = src[b++];
                else dst[k++] = src[a++];
            }
            while (a < mid) dst[k++] = src[a++];
            while (b < hi) dst[k++] = src[b++];
        }
        int64_t *t = src; src = dst; dst = t;
    }
    if (src != o->perm) memcpy(o->perm, src, sizeof(int64_t) * (size_t)N);
}

static void update_neighbors(orc_handle *o) {
    const int D = o->D;
    const int64_t N = o->N;
    for (int64_t i = 0; i < N; ++i)
        for (int d = 0; d < D; ++d) o->cells[i * D + d] = map_floor(o->pos[i * D + d], o->cfg.H_inv);
    int sorted = 1;
    for (int64_t i = 1; i < N && sorted; ++i)
        if (cell_cmp(&o->cells[i * D], &o->cells[(i - 1) * D], D) < 0) sorted = 0;
    if (!sorted) {
        for (int64_t i = 0; i < N; ++i) o->perm[i] = i;
        merge_sort_perm(o);
        /* sort! permutes all 17 StructArray fields (src/SPHCellList.jl:142, src/PreProcess.jl:114) */
        permute_i64(o, o->cells, D);
        permute_f64(o, o->pos, D);
        permute_f64(o, o->acc, D);
        permute_f64(o, o->vel, D);
        permute_f64(o, o->rho, 1);
        permute_f64(o, o->press, 1);
        permute_f64(o, o->gf, 1);
        permute_f64(o, o->ml, 1);
        permute_f64(o, o->ghost, D);
        permute_i64(o, o->id, 1);
        permute_i64(o, o->prow, 1);
        permute_i64(o, (int64_t *)o->group, 1);
        permute_u8(o, o->type);
    }
    /* :144-160 — ranges are kept as the reference's 1-based particle indices */
    memset(o->ranges, 0, sizeof(int64_t) * (size_t)(N + 2));
    o->ranges[0] = 1;                 /* ParticleRanges[1] = 1 (dummy empty cell)              */
    int64_t ic = 2;                   /* IndexCounter                                            */
    o->ranges[ic - 1] = 1;
    memcpy(&o->ucells[(ic - 1) * D], &o->cells[0], sizeof(int64_t) * D);
    for (int64_t i = 1; i < N; ++i) {
        if (cell_cmp(&o->cells[i * D], &o->cells[(i - 1) * D], D) != 0) {
            ic += 1;
            o->ranges[ic - 1] = i + 1;
            memcpy(&o->ucells[(ic - 1) * D], &o->cells[i * D], sizeof(int64_t) * D);
        }
    }
    o->ranges[ic] = N + 1;            /* deviation Q5: reference writes N+2 here                 */
    o->index_counter = ic;
    o->n_rebuilds += 1;
}

/* get(CellDict, cell, 1): slot (1-based) of `cell`, 1 (dummy) when absent — src/SPHCellList.jl:201.
 * The unique cells are in sort order, so a binary search stands in for the Dict. */
static int64_t cell_slot(const orc_handle *o, const int64_t *cell) {
    const int D = o->D;
    int64_t lo = 1, hi = o->index_counter - 1; /* 0-based slots 1..ic-1 hold real cells */
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        int c = cell_cmp(&o->ucells[mid * D], cell, D);
        if (c == 0) return mid + 1;
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return 1;
}

/* SimulationEquations.jl:49-62 Estimate7thRoot / InverseHydrostaticEquationOfState (Float64 bit trick + two
 * Newton steps of t³ − x/t⁴ = 0; @fastmath may re-associate: not reproduced) */
static inline double estimate_7th_root(double x) {
    union { double d; uint64_t u; } a, t0;
    a.d = fabs(x);
    t0.u = 0x36cd000000000000ull + a.u / 7ull;
    double t = copysign(t0.d, x);
    for (int k = 0; k < 2; ++k) {
        double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, xot4 = x / t4;
        t = t - t * (t3 - xot4) / (4.0 * t3 + 3.0 * xot4);
    }
    return t;
}

/* ---------------------------------------------------------------------------------------------
 * SPHCellList.jl:268-317 ComputeInteractions!  with the density-diffusion models of
 * SPHDensityDiffusionModels.jl:30-188, the viscosity models of SPHViscosityModels.jl:51-126 and
 * add_shifting_terms! (SPHCellList.jl:73-88).
 * i, j are 0-based.  `rho` / `vel` are the Density / Velocity arguments of the pair loop;
 * o->rho / o->vel are SimParticles.Density / .Velocity (quirk Q2: several model terms read those).
 * ------------------------------------------------------------------------------------------- */
static inline void compute_interactions(const orc_handle *o, const double *pos, const double *rho,
                                        const double *press, const double *vel,
                                        int64_t i, int64_t j, double *drho_t, double *acc_t,
                                        double *gradC_t, double *divr_t, double *kern_t, double *kgrad_t) {
    const sphmi_config *c = &o->cfg;
    const int D = o->D;
    double xij[MAXD], r2 = 0.0;
    for (int d = 0; d < D; ++d) {
        xij[d] = pos[i * D + d] - pos[j * D + d];
        r2 += xij[d] * xij[d];
    }
    if (!(r2 <= c->H2)) return;
    double dij = sqrt(fabs(r2));
    double q = dij * c->h_inv;
    q = q < 0.0 ? 0.0 : (q > 2.0 ? 2.0 : q);
    double fac = gradW_factor_r(c, q, dij);
    double gW[MAXD];
    for (int d = 0; d < D; ++d) gW[d] = fac * xij[d];

    double rho_i = rho[i], rho_j = rho[j];
    double vij[MAXD], sym = 0.0;
    for (int d = 0; d < D; ++d) {
        vij[d] = vel[i * D + d] - vel[j * D + d];
        sym += -vij[d] * gW[d];
    }
    double drho_p = -rho_i * (c->m0 / rho_j) * sym;
    double drho_m = -rho_j * (c->m0 / rho_i) * sym;

    /* density diffusion — reads SimParticles.Density (quirk Q2) */
    double Di = 0.0, Dj = 0.0;
    if (c->density_diffusion != SPHMI_DDT_NONE) {
        double rn_i = o->rho[i], rn_j = o->rho[j];
        double rhoH = 0.0, mlc = 1.0;
        if (c->density_diffusion == SPHMI_DDT_LINEAR) {                       /* :100-136 */
            double lin = (1.0 / (c->Cb * c->gamma)) * c->rho0;
            double PH = c->rho0 * (-c->g) * -xij[D - 1];
            rhoH = PH * lin;
            mlc = o->ml[i] * o->ml[j];
        } else if (c->density_diffusion == SPHMI_DDT_COMPLEX) {               /* :150-188 */
            double PH = c->rho0 * (-c->g) * -xij[D - 1];
            rhoH = c->rho0 * (estimate_7th_root(1.0 + PH * (1.0 / c->Cb)) - 1.0);
            mlc = o->ml[i] * o->ml[j];
        }                                                                     /* :56-87: no ρᴴ, no MLcond */
        double inv = 1.0 / (r2 + c->eta2);
        double rji = rn_j - rn_i;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += (2.0 * (rji - rhoH) * (-xij[d]) * inv) * gW[d];
        Di = c->delta_phi * c->h * c->c0 * (c->m0 / rn_j) * dot * mlc;
        Dj = -Di;
    }
    drho_t[i] += drho_p + Di;
    drho_t[j] += drho_m + Dj;

    double Pfac = (press[i] + press[j]) / (rho_i * rho_j);
    double um[MAXD];
    double f_ab = tensile_correction(c, press[i], rho_i, press[j], rho_j, q);
    for (int d = 0; d < D; ++d) um[d] = -c->m0 * (Pfac + f_ab) * gW[d];

    if (c->viscosity == SPHMI_VISC_ARTIFICIAL) {                              /* :56-74 */
        double rn_i = o->rho[i], rn_j = o->rho[j];
        double vdx = 0.0;
        for (int d = 0; d < D; ++d) vdx += vij[d] * xij[d];
        if (vdx < 0.0) {
            double rbar = 0.5 * (rn_i + rn_j);
            double mu = c->h * vdx / (r2 + c->eta2);
            double k = -c->m0 * (-c->alpha * c->c0 * mu) / rbar;
            for (int d = 0; d < D; ++d) um[d] += k * gW[d];
        }
    } else if (c->viscosity == SPHMI_VISC_LAMINAR || c->viscosity == SPHMI_VISC_LAMINAR_SPS) {
        double rn_i = o->rho[i], rn_j = o->rho[j];
        double xdg = 0.0;
        for (int d = 0; d < D; ++d) xdg += xij[d] * gW[d];
        /* :77-87 — the denominator adds (ρᵢ+ρⱼ) and (d²+η²) exactly as the reference does */
        double term = (4.0 * c->m0 * c->nu0 * xdg) / ((rn_i + rn_j) + (r2 + c->eta2));
        for (int d = 0; d < D; ++d) um[d] += term * vij[d];
        if (c->viscosity == SPHMI_VISC_LAMINAR_SPS) {                         /* :90-126 */
            const double *vi = &o->vel[i * D], *vj = &o->vel[j * D];          /* SimParticles.Velocity */
            double Si[MAXD][MAXD], Sj[MAXD][MAXD], ti[MAXD][MAXD], tj[MAXD][MAXD];
            double ssi = 0.0, ssj = 0.0, tri = 0.0, trj = 0.0;
            for (int r = 0; r < D; ++r)
                for (int cc = 0; cc < D; ++cc) {
                    Si[r][cc] = (c->m0 / rn_j) * (vj[r] - vi[r]) * gW[cc];
                    Sj[r][cc] = (c->m0 / rn_i) * (vi[r] - vj[r]) * -gW[cc];
                    ssi += Si[r][cc] * Si[r][cc];
                    ssj += Sj[r][cc] * Sj[r][cc];
                    if (r == cc) { tri += Si[r][cc]; trj += Sj[r][cc]; }
                }
            double ni = sqrt(2.0 * ssi), nj = sqrt(2.0 * ssj);
            double csdx2 = (c->smagorinsky_constant * c->dx) * (c->smagorinsky_constant * c->dx);
            double nuti = csdx2 * ni, nutj = csdx2 * nj;
            for (int r = 0; r < D; ++r)
                for (int cc = 0; cc < D; ++cc) {
                    double I = r == cc ? 1.0 : 0.0;
                    ti[r][cc] = 2.0 * nuti * rn_i * (Si[r][cc] - (1.0 / 3.0) * tri * I)
                                - (2.0 / 3.0) * rn_i * c->blin_constant * (c->dx * c->dx) * (ni * ni) * I;
                    tj[r][cc] = 2.0 * nutj * rn_j * (Sj[r][cc] - (1.0 / 3.0) * trj * I)
                                - (2.0 / 3.0) * rn_j * c->blin_constant * (c->dx * c->dx) * (nj * nj) * I;
                }
            for (int r = 0; r < D; ++r) {
                double sacc = 0.0;
                for (int cc = 0; cc < D; ++cc) sacc += (ti[r][cc] + tj[r][cc]) * gW[cc];
                um[r] += (c->m0 / (rn_j * rn_i)) * sacc;
            }
        }
    }
    for (int d = 0; d < D; ++d) {
        acc_t[i * D + d] += um[d];
        acc_t[j * D + d] -= um[d];
    }
    if (kern_t) {                                                             /* KernelOutput!, :106-116 */
        double Wij = W_kernel(c, q);
        kern_t[i] += Wij; kern_t[j] += Wij;
        for (int d = 0; d < D; ++d) { kgrad_t[i * D + d] += gW[d]; kgrad_t[j * D + d] -= gW[d]; }
    }
    if (gradC_t) {                                                            /* add_shifting_terms!, :73-88 */
        double mlc = o->ml[i] * o->ml[j];
        double xdg = 0.0;
        for (int d = 0; d < D; ++d) xdg += xij[d] * gW[d];
        for (int d = 0; d < D; ++d) {
            gradC_t[i * D + d] += (c->m0 / rho_i) * gW[d];
            gradC_t[j * D + d] += (c->m0 / rho_j) * -gW[d];
        }
        divr_t[i] += (c->m0 / rho_j) * (-xdg) * mlc;
        divr_t[j] += (c->m0 / rho_i) * (-xdg) * mlc;
    }
}

/* SPHCellList.jl:37-43 ConstructStencil: first half (column-major) of the 3^D offsets */
static int construct_stencil(int D, int64_t st[][MAXD]) {
    int total = 1;
    for (int d = 0; d < D; ++d) total *= 3;
    int half = total / 2;
    for (int s = 0; s < half; ++s) {
        int r = s;
        for (int d = 0; d < D; ++d) { st[s][d] = (r % 3) - 1; r /= 3; }
    }
    return half;
}

/* SPHCellList.jl:168-217 NeighborLoop!  + :416-432 ResetStep! + :476-484 ReductionStep! */
static void neighbor_loop(orc_handle *o, const double *pos, const double *rho, const double *press,
                          const double *vel) {
    const int D = o->D;
    const int64_t N = o->N;
    const int T = o->nthreads;
    int64_t stencil[13][MAXD];
    const int ns = construct_stencil(D, stencil);
    const int64_t ncell_dict = o->index_counter - 1;      /* length(CellDict)          */
    const int64_t nview = o->index_counter;               /* length(UniqueCellsView)   */
    int64_t base = (ncell_dict + T - 1) / T;
    int64_t chunk = base + (base & 1);
    if (chunk < 1) chunk = 1;
    int64_t nchunks = (nview + chunk - 1) / chunk;

    const int shift = o->cfg.shifting == SPHMI_SHIFT_PLANAR;
    const int kout = o->cfg.kernel_output == SPHMI_KOUT_STORE;
    /* ResetStep! */
    memset(o->drhodt, 0, sizeof(double) * (size_t)N);
    memset(o->acc, 0, sizeof(double) * (size_t)N * D);
    if (shift) { memset(o->gradC, 0, sizeof(double) * (size_t)N * D); memset(o->divr, 0, sizeof(double) * (size_t)N); }
    if (kout) { memset(o->kern, 0, sizeof(double) * (size_t)N); memset(o->kgrad, 0, sizeof(double) * (size_t)N * D); }
#pragma omp parallel for num_threads(T) schedule(static)
    for (int t = 0; t < T; ++t) {
        memset(o->drhodt_thr[t], 0, sizeof(double) * (size_t)N);
        memset(o->acc_thr[t], 0, sizeof(double) * (size_t)N * D);
        if (shift) { memset(o->gradC_thr[t], 0, sizeof(double) * (size_t)N * D); memset(o->divr_thr[t], 0, sizeof(double) * (size_t)N); }
        if (kout) { memset(o->kern_thr[t], 0, sizeof(double) * (size_t)N); memset(o->kgrad_thr[t], 0, sizeof(double) * (size_t)N * D); }
    }

#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        int copy = (int)(ch % T);
        /* chunks ch, ch+T, ... share a copy; with schedule(static,1) they also share a thread */
        double *drho_t = o->drhodt_thr[copy];
        double *acc_t = o->acc_thr[copy];
        double *gC_t = shift ? o->gradC_thr[copy] : NULL, *dr_t = shift ? o->divr_thr[copy] : NULL;
        double *kn_t = kout ? o->kern_thr[copy] : NULL, *kg_t = kout ? o->kgrad_thr[copy] : NULL;
        int64_t it0 = ch * chunk + 1, it1 = it0 + chunk - 1;  /* 1-based iter as in the reference */
        if (it1 > nview) it1 = nview;
        for (int64_t iter = it0; iter <= it1; ++iter) {
            const int64_t *cell = &o->ucells[(iter - 1) * D];
            int64_t s0 = o->ranges[iter - 1], e0 = o->ranges[iter] - 1;  /* 1-based inclusive */
            for (int64_t i = s0; i <= e0; ++i)
                for (int64_t j = i + 1; j <= e0; ++j)
                    compute_interactions(o, pos, rho, press, vel, i - 1, j - 1, drho_t, acc_t, gC_t, dr_t, kn_t, kg_t);
            for (int s = 0; s < ns; ++s) {
                int64_t sc[MAXD];
                for (int d = 0; d < D; ++d) sc[d] = cell[d] + stencil[s][d];
                int64_t nb = cell_slot(o, sc);
                int64_t s1 = o->ranges[nb - 1], e1 = o->ranges[nb] - 1;
                for (int64_t i = s0; i <= e0; ++i)
                    for (int64_t j = s1; j <= e1; ++j)
                        compute_interactions(o, pos, rho, press, vel, i - 1, j - 1, drho_t, acc_t, gC_t, dr_t, kn_t, kg_t);
            }
        }
    }
    /* reduce_sum! (:367-381): target[i] += copy_j[i] for j = 1..nthreads in order */
#pragma omp parallel for num_threads(T) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        for (int t = 0; t < T; ++t) {
            o->drhodt[i] += o->drhodt_thr[t][i];
            for (int d = 0; d < D; ++d) o->acc[i * D + d] += o->acc_thr[t][i * D + d];
            if (kout) {
                o->kern[i] += o->kern_thr[t][i];
                for (int d = 0; d < D; ++d) o->kgrad[i * D + d] += o->kgrad_thr[t][i * D + d];
            }
            if (shift) {
                o->divr[i] += o->divr_thr[t][i];
                for (int d = 0; d < D; ++d) o->gradC[i * D + d] += o->gradC_thr[t][i * D + d];
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * mDBC: SPHCellList.jl:219-266 NeighborLoopMDBC!, :319-365 ComputeInteractionsMDBC!,
 *       :598-622 ApplyMDBCCorrection.
 * ------------------------------------------------------------------------------------------- */
static double det_n(const double *A, int n) {   /* A column-major n x n, n = 3 or 4 */
#define AA(r, c) A[(c) * n + (r)]
    if (n == 3) {
        return AA(0,0) * (AA(1,1) * AA(2,2) - AA(1,2) * AA(2,1))
             - AA(0,1) * (AA(1,0) * AA(2,2) - AA(1,2) * AA(2,0))
             + AA(0,2) * (AA(1,0) * AA(2,1) - AA(1,1) * AA(2,0));
    }
    double det = 0.0;
    for (int c = 0; c < 4; ++c) {
        double m[9];
        int cc = 0;
        for (int c2 = 0; c2 < 4; ++c2) {
            if (c2 == c) continue;
            for (int r = 1; r < 4; ++r) m[cc * 3 + (r - 1)] = AA(r, c2);
            cc++;
        }
        double d3 = det_n(m, 3);
        det += ((c & 1) ? -1.0 : 1.0) * AA(0, c) * d3;
    }
    return det;
#undef AA
}
/* A\b by Gaussian elimination with partial pivoting (StaticArrays' `\` for 3x3/4x4 is an LU /
 * closed-form solve; equal to rounding — SURVEY §8c "tolerance-level parity only"). */
static void solve_n(const double *A, const double *b, int n, double *x) {
    double M[4][5];
    for (int r = 0; r < n; ++r) {
        for (int c = 0; c < n; ++c) M[r][c] = A[c * n + r];
        M[r][n] = b[r];
    }
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r) if (fabs(M[r][k]) > fabs(M[p][k])) p = r;
        if (p != k) for (int c = 0; c <= n; ++c) { double t = M[k][c]; M[k][c] = M[p][c]; M[p][c] = t; }
        for (int r = k + 1; r < n; ++r) {
            double f = M[r][k] / M[k][k];
            for (int c = k; c <= n; ++c) M[r][c] -= f * M[k][c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = M[r][n];
        for (int c = r + 1; c < n; ++c) s -= M[r][c] * x[c];
        x[r] = s / M[r][r];
    }
}

static void apply_mdbc_before_half(orc_handle *o) {
    const sphmi_config *c = &o->cfg;
    const int D = o->D, P = D + 1;
    const int64_t N = o->N;
    int nfull = 1;
    for (int d = 0; d < D; ++d) nfull *= 3;
#pragma omp parallel for num_threads(o->nthreads) schedule(static)
    for (int64_t it = 0; it < N; ++it) {
        const double *g = &o->ghost[it * D];
        int nz = 0;
        for (int d = 0; d < D; ++d) nz |= (g[d] != 0.0);
        if (!nz) continue;
        double b[4] = {0, 0, 0, 0}, A[16];
        memset(A, 0, sizeof(A));
        int64_t gc[MAXD];
        for (int d = 0; d < D; ++d) gc[d] = map_floor(g[d], c->H_inv);
        for (int s = 0; s < nfull; ++s) {
            int64_t sc[MAXD];
            int r = s;
            for (int d = 0; d < D; ++d) { sc[d] = gc[d] + (r % 3) - 1; r /= 3; }
            int64_t nb = cell_slot(o, sc);
            int64_t s1 = o->ranges[nb - 1], e1 = o->ranges[nb] - 1;
            for (int64_t j1 = s1; j1 <= e1; ++j1) {
                int64_t j = j1 - 1;
                if (o->type[j] != SPHMI_FLUID) continue;
                double xij[MAXD], r2 = 0.0;
                for (int d = 0; d < D; ++d) { xij[d] = g[d] - o->pos[j * D + d]; r2 += xij[d] * xij[d]; }
                if (!(r2 <= c->H2)) continue;
                double q = sqrt(fabs(r2)) * c->h_inv;
                q = q < 0.0 ? 0.0 : (q > 2.0 ? 2.0 : q);
                double Wij = W_wendland(c, q);
                double fac = gradW_factor(c, q);
                double Vj = c->m0 / o->rho[j];
                double fc[4];
                fc[0] = Vj * Wij;
                b[0] += c->m0 * Wij;
                for (int d = 0; d < D; ++d) {
                    double gw = fac * xij[d];
                    fc[d + 1] = Vj * gw;
                    b[d + 1] += c->m0 * gw;
                }
                for (int r2i = 0; r2i < P; ++r2i) {
                    A[0 * P + r2i] += fc[r2i];
                    for (int k = 0; k < D; ++k) A[(k + 1) * P + r2i] += (-xij[k]) * fc[r2i];
                }
            }
        }
        memcpy(&o->bgam[it * 4], b, sizeof(b));
        memcpy(&o->Agam[it * 16], A, sizeof(A));
    }
    /* ApplyMDBCCorrection */
    for (int64_t i = 0; i < N; ++i) {
        const double *g = &o->ghost[i * D];
        int nz = 0;
        for (int d = 0; d < D; ++d) nz |= (g[d] != 0.0);
        if (!nz) continue;
        const double *A = &o->Agam[i * 16];
        const double *b = &o->bgam[i * 4];
        if (fabs(det_n(A, P)) >= 1e-3) {
            double s[4];
            solve_n(A, b, P, s);
            double v1 = s[0];
            for (int d = 0; d < D; ++d) v1 += s[d + 1] * (o->pos[i * D + d] - g[d]);
            o->rho[i] = isnan(v1) ? c->rho0 : v1;
        } else if (A[0] > 0.0) {
            double v = b[0] / A[0];
            o->rho[i] = isnan(v) ? c->rho0 : v;
        }
    }
}

/* SPHCellList.jl:624-638 HalfTimeStep */
static void half_time_step(orc_handle *o, double dt2) {
    const int D = o->D;
    for (int64_t i = 0; i < o->N; ++i) {
        o->acc[i * D + D - 1] += o->cfg.g * o->gf[i];
        for (int d = 0; d < D; ++d) {
            o->pos_np[i * D + d] = o->pos[i * D + d] + o->vel[i * D + d] * dt2 * o->ml[i];
            o->vel_np[i * D + d] = o->vel[i * D + d] + o->acc[i * D + d] * dt2 * o->ml[i];
        }
        o->rho_np[i] = o->rho[i] + o->drhodt[i] * dt2;
    }
}

/* SPHCellList.jl:640-652 FullTimeStep (NoShifting) and :654-677 (PlanarShifting) */
static void full_time_step(orc_handle *o, double dt) {
    const int D = o->D;
    const int shift = o->cfg.shifting == SPHMI_SHIFT_PLANAR;
    const double A = 2.0, A_FST = 0.0, A_FSM = (double)D;
    for (int64_t i = 0; i < o->N; ++i) {
        o->acc[i * D + D - 1] += o->cfg.g * o->gf[i];
        for (int d = 0; d < D; ++d) o->vel[i * D + d] += o->acc[i * D + d] * dt * o->ml[i];
        double dxs[MAXD] = {0.0, 0.0, 0.0};
        if (shift) {
            double A_FSC = (o->divr[i] - A_FST) / (A_FSM - A_FST);
            if (!(A_FSC < 0.0)) {
                double vn = 0.0;
                for (int d = 0; d < D; ++d) vn += o->vel[i * D + d] * o->vel[i * D + d];
                vn = sqrt(vn);
                for (int d = 0; d < D; ++d) dxs[d] = -A_FSC * A * o->cfg.h * vn * dt * o->gradC[i * D + d];
            }
        }
        for (int d = 0; d < D; ++d) {
            double a = o->acc[i * D + d];
            double v = o->vel[i * D + d];
            o->pos[i * D + d] += (((v + (v - a * dt * o->ml[i])) / 2.0) * dt + dxs[d]) * o->ml[i];
        }
    }
}

/* SPHCellList.jl:575-596 ProgressMotion */
static void progress_motion(orc_handle *o, double dt2) {
    const int D = o->D;
    if (o->n_motion == 0) return;
    for (int64_t i = 0; i < o->N; ++i) {
        if (o->type[i] != SPHMI_MOVING) continue;
        for (int m = 0; m < o->n_motion; ++m) {
            if (o->motion[m].group != o->group[i]) continue;
            double should = (o->motion[m].start <= o->total_time &&
                             o->total_time <= o->motion[m].start + o->motion[m].dur) ? 1.0 : 0.0;
            for (int d = 0; d < D; ++d) {
                o->vel[i * D + d] = o->motion[m].vel * o->motion[m].dir[d] * should;
                o->pos[i * D + d] += o->vel[i * D + d] * dt2;
            }
            break;
        }
    }
}

/* SPHCellList.jl:742-802: one iteration of the while loop in SimulationLoop */
static int one_step(orc_handle *o) {
    o->delta_x = update_delta_x(o, o->delta_x);                 /* :744 */
    double dt = delta_t(o);                                     /* :748 */
    double dt2 = dt * 0.5;
    if (!(dt > 0.0) || isnan(dt)) {
        snprintf(o->err, sizeof(o->err), "non-positive or NaN dt at iteration %lld", (long long)o->iteration);
        return SPHMI_ERR_NUMERIC;
    }
    if (o->delta_x >= o->cfg.h) {                               /* :758-762 */
        update_neighbors(o);
        o->delta_x = 0.0;
    }
    progress_motion(o, dt2);                                    /* :765 */
    pressure(o, o->press, o->rho);                              /* :771 */
    if (o->cfg.mdbc == SPHMI_MDBC_SIMPLE) apply_mdbc_before_half(o);   /* :772 */
    neighbor_loop(o, o->pos, o->rho, o->press, o->vel);         /* :768,:774-775 */
    half_time_step(o, dt2);                                     /* :778 */
    limit_density_at_boundary(o, o->rho_np);                    /* :781 */
    progress_motion(o, dt2);                                    /* :787 */
    pressure(o, o->press, o->rho_np);                           /* :789 */
    neighbor_loop(o, o->pos_np, o->rho_np, o->press, o->vel_np);/* :784,:790-791 */
    limit_density_at_boundary(o, o->rho);                       /* :794 */
    density_epsi(o, dt);                                        /* :796 */
    full_time_step(o, dt);                                      /* :798 */
    o->iteration += 1;                                          /* :800, :679-685 */
    o->last_dt = dt;
    o->total_time += dt;
    return SPHMI_OK;
}

/* ============================================================================================
 * C API (same shape as include/sphmi.h so tests drive oracle and engine identically)
 * ========================================================================================== */
static char g_err[256];

const char *orc_last_error(const orc_handle *o) { return o ? o->err : g_err; }

int orc_set_threads(orc_handle *o, int n);

static void *xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

int orc_create(const sphmi_config *cfg, orc_handle **out) {
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(sphmi_config) ||
        (cfg->dims != 2 && cfg->dims != 3) || cfg->n_particles < 1 || cfg->host_float_bytes != 8) {
        snprintf(g_err, sizeof(g_err), "orc_create: bad config (oracle takes fp64 host arrays)");
        return SPHMI_ERR_ARGUMENT;
    }
    orc_handle *o = (orc_handle *)calloc(1, sizeof(orc_handle));
    o->cfg = *cfg;
    o->D = cfg->dims;
    o->N = cfg->n_particles;
    const size_t N = (size_t)o->N, D = (size_t)o->D;
    o->pos = xcalloc(N * D, 8); o->vel = xcalloc(N * D, 8); o->acc = xcalloc(N * D, 8);
    o->rho = xcalloc(N, 8); o->press = xcalloc(N, 8); o->gf = xcalloc(N, 8); o->ml = xcalloc(N, 8);
    o->ghost = xcalloc(N * D, 8); o->type = xcalloc(N, 1); o->id = xcalloc(N, 8); o->prow = xcalloc(N, 8); o->group = xcalloc(N, 8);
    o->cells = xcalloc(N * D, 8);
    o->drhodt = xcalloc(N, 8); o->vel_np = xcalloc(N * D, 8); o->pos_np = xcalloc(N * D, 8); o->rho_np = xcalloc(N, 8);
    o->ranges = xcalloc(N + 2, 8); o->ucells = xcalloc((N + 1) * D, 8);
    o->perm = xcalloc(N, 8); o->perm_tmp = xcalloc(N, 8); o->scratch = xcalloc(N * D, 8);
    o->bgam = xcalloc(N * 4, 8); o->Agam = xcalloc(N * 16, 8);
    o->gradC = xcalloc(N * D, 8); o->divr = xcalloc(N, 8);
    o->kern = xcalloc(N, 8); o->kgrad = xcalloc(N * D, 8);
    o->nthreads = 0;
    orc_set_threads(o, 1);
    *out = o;
    return SPHMI_OK;
}

int orc_set_threads(orc_handle *o, int n) {
    if (n < 1) n = 1;
    if (o->drhodt_thr) {
        for (int t = 0; t < o->nthreads; ++t) { free(o->drhodt_thr[t]); free(o->acc_thr[t]); free(o->gradC_thr[t]); free(o->divr_thr[t]); free(o->kern_thr[t]); free(o->kgrad_thr[t]); }
        free(o->drhodt_thr); free(o->acc_thr); free(o->gradC_thr); free(o->divr_thr); free(o->kern_thr); free(o->kgrad_thr);
    }
    o->nthreads = n;
    o->drhodt_thr = calloc((size_t)n, sizeof(double *));
    o->acc_thr = calloc((size_t)n, sizeof(double *));
    o->gradC_thr = calloc((size_t)n, sizeof(double *));
    o->divr_thr = calloc((size_t)n, sizeof(double *));
    o->kern_thr = calloc((size_t)n, sizeof(double *));
    o->kgrad_thr = calloc((size_t)n, sizeof(double *));
    const int kout = o->cfg.kernel_output == SPHMI_KOUT_STORE;
    const int shift = o->cfg.shifting == SPHMI_SHIFT_PLANAR;
    for (int t = 0; t < n; ++t) {
        o->drhodt_thr[t] = xcalloc((size_t)o->N, 8);
        o->acc_thr[t] = xcalloc((size_t)o->N * o->D, 8);
        o->gradC_thr[t] = xcalloc(shift ? (size_t)o->N * o->D : 1, 8);
        o->divr_thr[t] = xcalloc(shift ? (size_t)o->N : 1, 8);
        o->kern_thr[t] = xcalloc(kout ? (size_t)o->N : 1, 8);
        o->kgrad_thr[t] = xcalloc(kout ? (size_t)o->N * o->D : 1, 8);
    }
    return SPHMI_OK;
}

int orc_destroy(orc_handle *o) {
    if (!o) return SPHMI_OK;
    for (int t = 0; t < o->nthreads; ++t) { free(o->drhodt_thr[t]); free(o->acc_thr[t]); free(o->gradC_thr[t]); free(o->divr_thr[t]); free(o->kern_thr[t]); free(o->kgrad_thr[t]); }
    free(o->drhodt_thr); free(o->acc_thr); free(o->gradC_thr); free(o->divr_thr); free(o->kern_thr); free(o->kgrad_thr);
    free(o->gradC); free(o->divr); free(o->kern); free(o->kgrad);
    free(o->pos); free(o->vel); free(o->acc); free(o->rho); free(o->press); free(o->gf); free(o->ml);
    free(o->ghost); free(o->type); free(o->id); free(o->prow); free(o->group); free(o->cells);
    free(o->drhodt); free(o->vel_np); free(o->pos_np); free(o->rho_np);
    free(o->ranges); free(o->ucells); free(o->perm); free(o->perm_tmp); free(o->scratch);
    free(o->bgam); free(o->Agam);
    free(o);
    return SPHMI_OK;
}

int orc_upload(orc_handle *o, const double *position, const double *velocity, const double *acceleration,
               const double *density, const uint8_t *type, const int64_t *id, const uint64_t *group,
               const double *ghost_points) {
    if (!o || !position || !velocity || !density || !type || !id) return SPHMI_ERR_ARGUMENT;
    const size_t N = (size_t)o->N, D = (size_t)o->D;
    memcpy(o->pos, position, N * D * 8);
    memcpy(o->vel, velocity, N * D * 8);
    if (acceleration) memcpy(o->acc, acceleration, N * D * 8); else memset(o->acc, 0, N * D * 8);
    memcpy(o->rho, density, N * 8);
    memcpy(o->type, type, N);
    memcpy(o->id, id, N * 8);
    for (int64_t i = 0; i < (int64_t)N; ++i) o->prow[i] = i;
    if (group) memcpy(o->group, group, N * 8); else memset(o->group, 0, N * 8);
    if (ghost_points) memcpy(o->ghost, ghost_points, N * D * 8); else memset(o->ghost, 0, N * D * 8);
    /* src/PreProcess.jl:78-98 */
    for (size_t i = 0; i < N; ++i) {
        o->gf[i] = type[i] == SPHMI_FLUID ? -1.0 : (type[i] == SPHMI_MOVING ? 1.0 : 0.0);
        o->ml[i] = type[i] == SPHMI_FLUID ? 1.0 : 0.0;
    }
    memset(o->pos_np, 0, N * D * 8);
    memset(o->vel_np, 0, N * D * 8);
    memset(o->rho_np, 0, N * 8);
    memset(o->drhodt, 0, N * 8);
    pressure(o, o->press, o->rho);         /* src/SPHCellList.jl:835 */
    o->index_counter = 0;
    o->uploaded = 1;
    return SPHMI_OK;
}

/* The sort as a permutation (include/sphmi.h: sphmi_download_permutation): the reference's sort! (src/SPHCellList.jl:142)
 * permutes every column of the StructArray; `prow` is a row-number column that rides along. */
int orc_download_permutation(orc_handle *o, int64_t *prev_row) {
    if (!o || !prev_row) return SPHMI_ERR_ARGUMENT;
    for (int64_t i = 0; i < o->N; ++i) { prev_row[i] = o->prow[i]; o->prow[i] = i; }
    return SPHMI_OK;
}

int orc_download_kernel_output(orc_handle *o, double *kernel, double *kernel_gradient) {
    if (!o || o->cfg.kernel_output != SPHMI_KOUT_STORE) return SPHMI_ERR_STATE;
    if (kernel) memcpy(kernel, o->kern, (size_t)o->N * 8);
    if (kernel_gradient) memcpy(kernel_gradient, o->kgrad, (size_t)o->N * o->D * 8);
    return SPHMI_OK;
}

int orc_set_motion(orc_handle *o, uint64_t group, double velocity, double start_time, double duration,
                   const double *direction) {
    if (!o || !direction) return SPHMI_ERR_ARGUMENT;
    int m = 0;
    while (m < o->n_motion && o->motion[m].group != group) ++m;
    if (m == 16) { snprintf(o->err, sizeof(o->err), "orc_set_motion: more than 16 groups"); return SPHMI_ERR_ARGUMENT; }
    if (m == o->n_motion) o->n_motion += 1;
    o->motion[m].group = group; o->motion[m].vel = velocity; o->motion[m].start = start_time; o->motion[m].dur = duration;
    for (int d = 0; d < MAXD; ++d) o->motion[m].dir[d] = d < o->D ? direction[d] : 0.0;
    return SPHMI_OK;
}

int orc_set_clock(orc_handle *o, int64_t iteration, double total_time) {
    o->iteration = iteration;
    o->total_time = total_time;
    return SPHMI_OK;
}

static void fill_progress(orc_handle *o, int64_t steps, sphmi_progress *p) {
    if (!p) return;
    p->iteration = o->iteration; p->steps_done = steps; p->n_rebuilds = o->n_rebuilds;
    p->index_counter = o->index_counter; p->total_time = o->total_time; p->last_dt = o->last_dt;
    p->delta_x = o->delta_x;
}

int orc_advance(orc_handle *o, double t_target, int64_t max_steps, sphmi_progress *out) {
    if (!o || !o->uploaded) return SPHMI_ERR_STATE;
    o->delta_x = 1.0 + o->cfg.h;                                 /* src/SPHCellList.jl:739 */
    int64_t steps = 0;
    while (o->total_time <= t_target && (max_steps < 0 || steps < max_steps)) {
        int rc = one_step(o);
        if (rc != SPHMI_OK) { fill_progress(o, steps, out); return rc; }
        steps++;
    }
    fill_progress(o, steps, out);
    return SPHMI_OK;
}

int orc_download(orc_handle *o, double *position, double *velocity, double *acceleration, double *density,
                 double *pressure_out, int64_t *id, uint8_t *type, uint64_t *group, double *ghost_points,
                 int64_t *cells) {
    const size_t N = (size_t)o->N, D = (size_t)o->D;
    if (position) memcpy(position, o->pos, N * D * 8);
    if (velocity) memcpy(velocity, o->vel, N * D * 8);
    if (acceleration) memcpy(acceleration, o->acc, N * D * 8);
    if (density) memcpy(density, o->rho, N * 8);
    if (pressure_out) memcpy(pressure_out, o->press, N * 8);
    if (id) memcpy(id, o->id, N * 8);
    if (type) memcpy(type, o->type, N);
    if (group) memcpy(group, o->group, N * 8);
    if (ghost_points) memcpy(ghost_points, o->ghost, N * D * 8);
    if (cells) memcpy(cells, o->cells, N * D * 8);
    return SPHMI_OK;
}

int orc_forces_once(orc_handle *o, int apply_mdbc, double *drhodt, double *acceleration) {
    if (!o || !o->uploaded) return SPHMI_ERR_STATE;
    update_neighbors(o);
    pressure(o, o->press, o->rho);
    if (apply_mdbc && o->cfg.mdbc == SPHMI_MDBC_SIMPLE) apply_mdbc_before_half(o);
    neighbor_loop(o, o->pos, o->rho, o->press, o->vel);
    if (drhodt) memcpy(drhodt, o->drhodt, (size_t)o->N * 8);
    if (acceleration) memcpy(acceleration, o->acc, (size_t)o->N * o->D * 8);
    return SPHMI_OK;
}

int orc_unique_cells(orc_handle *o, int64_t *cells_out, int64_t capacity, int64_t *n_out) {
    int64_t n = o->index_counter > 0 ? o->index_counter - 1 : 0;
    if (n_out) *n_out = n;
    if (cells_out) {
        if (capacity < n) return SPHMI_ERR_ARGUMENT;
        memcpy(cells_out, &o->ucells[o->D], (size_t)n * o->D * 8);
    }
    return SPHMI_OK;
}

/* ρₙ⁺ of the LAST step taken — the half-step density after LimitDensityAtBoundary! (src/SPHCellList.jl:778-781), which the reference feeds to
 * Pressure! and the second NeighborLoop! and then forgets.  The reference carries on with a non-positive value there; an engine that keeps the
 * MotionLimiter flag in the sign of ρ must refuse such a step — the tests ask the oracle whether that is what happened. */
int orc_half_step_density(orc_handle *o, double *out) {
    if (!o || !o->uploaded || !out) return SPHMI_ERR_STATE;
    memcpy(out, o->rho_np, (size_t)o->N * 8);
    return SPHMI_OK;
}

/* Stand-alone Δt for the upstream "time stepping" test (test/runtests.jl:6-16). */
double orc_delta_t(orc_handle *o) { return delta_t(o); }

/* Upstream "isolated particle" sequence (test/runtests.jl:52-66): one manual step without any
 * neighbour loop — Δt, HalfTimeStep, LimitDensityAtBoundary!, Pressure!, FullTimeStep,
 * DensityEpsi!, LimitDensityAtBoundary!, UpdateMetaData!  (ResetArrays! zeroes dρdtI and acc first). */
int orc_isolated_step(orc_handle *o) {
    memset(o->drhodt, 0, sizeof(double) * (size_t)o->N);
    memset(o->acc, 0, sizeof(double) * (size_t)o->N * o->D);
    double dt = delta_t(o);
    half_time_step(o, dt / 2);
    limit_density_at_boundary(o, o->rho_np);
    pressure(o, o->press, o->rho_np);
    full_time_step(o, dt);
    density_epsi(o, dt);
    limit_density_at_boundary(o, o->rho);
    o->iteration += 1; o->last_dt = dt; o->total_time += dt;
    return SPHMI_OK;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
