/*
 * sphmi_internal.h — test hooks of libsphmi.so's slab driver.  NOT part of the drop-in boundary (include/sphmi.h): nothing
 * here is bound by the Julia shim or needed to run a simulation.  The tests use them to start from chosen cuts, to check
 * the host-side planner and the device-side work measure against independent Python code, and to exercise the
 * shared-memory transport between CPU processes.
 */
#ifndef SPHMI_INTERNAL_H
#define SPHMI_INTERNAL_H

#include "sphmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* No device needed: rank `rank` of `world` processes attaches to the shared-memory transport keyed by `unique_id`
 * (128 bytes, the same in every process), runs SUM / MAX reductions of known vectors and a neighbour exchange of
 * `n_bytes` per direction with known patterns, and checks what arrives.  0 = everything matched. */
int sphmi_shm_selftest(const void* unique_id, int32_t rank, int32_t world, int64_t n_bytes);

/* Start from these cuts (world-1 first columns) instead of the balanced ones; call before sphmi_upload. */
int sphmi_multi_set_cuts(sphmi_handle* h, const int64_t* cuts, int32_t n);

/* Host-only planning (no device needed): slab axis, cuts, halo width, owned particles and capacity per slab for `world`
 * slabs of the given particle set.  cuts_out: world-1, owned_out / capacity_out: world entries. */
int sphmi_plan_slabs(const sphmi_config* cfg, const void* position, const void* ghost_points, int64_t n, int32_t world,
                     int32_t* axis_out, int32_t* halo_width_out, int64_t* cuts_out, int64_t* owned_out,
                     int64_t* capacity_out);

/* The work measure of the re-cut, as the slabs of a multi-device handle compute it on their devices after a rebuild:
 * cost_out[c] = Σ over the OWNED particles of cell column col0 + c (along the slab axis) of the candidates in their 3^D
 * cells, summed over the local slabs.  cost_out: ncols uint64 on the HOST. */
int sphmi_multi_column_cost(sphmi_handle* h, int64_t col0, int32_t ncols, uint64_t* cost_out);

/* What the scaling prediction of DESIGN.md section 7 is computed from (tools/scaling_inputs.py): twelve words per local slab of a multi-device
 * handle — { slab, rows held (owned + ghost copies), halo records sent to the left / right neighbour with state A, the same with the half-step
 * state H, tiles of the interior launch, tiles of the slab-edge launch, blocks per XCD run of the two launches, and — with
 * $SPHMI_DD_ONE_SLAB_AT_A_TIME=1, a measurement mode in which the slabs of a handle that share one GPU take their passes one after the other —
 * the mean nanoseconds of the slab's pass 1 / pass 2 (interior launch beside unpack + slab-edge launch) with the chip to itself }.  A halo record
 * is two packets (32 bytes with fp32 kernels, 64 with fp64). */
int sphmi_multi_halo_info(sphmi_handle* h, int64_t* out, int32_t capacity_words, int32_t* n_words_out);

#ifdef __cplusplus
}
#endif
#endif /* SPHMI_INTERNAL_H */
