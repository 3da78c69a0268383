/*
 * sphmi.h — C ABI of the MI355X-native SPH neighbour + force engine (libsphmi.so).
 *
 * This is the drop-in boundary for the hot path of AhmedSalih3d/SPHExample: one call to
 * sphmi_advance() replaces one call to the reference's
 *     SimulationLoop(...)                      src/SPHCellList.jl:727-805 (called at :883)
 * i.e. "advance the particle system until TotalTime > next_output_time", including the
 * cell-list rebuild (UpdateNeighbors!, :138-163), both neighbour passes
 * (NeighborLoop!/ComputeInteractions!, :168-217 / :268-317), mDBC (:219-266, :319-365, :598-622),
 * the predictor / corrector (HalfTimeStep :624-638, FullTimeStep :640-652, DensityEpsi!,
 * LimitDensityAtBoundary!, Pressure!  src/SimulationEquations.jl:9-42) and the adaptive time step
 * (src/TimeStepping.jl:24-46, update_delta_x! src/SPHCellList.jl:706-724).
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions cross the boundary; every function returns an int
 *     status (SPHMI_OK == 0) and sphmi_last_error() gives the text of the last failure.
 *   - the caller owns every host array it passes; pointers only need to stay valid for the
 *     duration of the call (Julia: GC.@preserve around the ccall).  The engine owns all device
 *     memory behind the opaque handle.
 *   - vector fields cross the boundary exactly as Julia lays out Vector{SVector{D,T}}:
 *     contiguous AoS, D components interleaved, N*D scalars.
 *   - host scalars are `host_float_bytes` wide (8 for every stock example), device arithmetic is
 *     `device_float_bytes` wide (4 = fp32 kernels, 8 = fp64 kernels, 0 = the library chooses:
 *     sphmi_auto_device_float_bytes); conversion happens in upload/download.
 *   - the library installs no signal handlers and never calls back into the host runtime.
 *   - one handle = one simulation; a handle must not be used from two threads at once.  A handle created with a
 *     device list (sphmi_config.n_devices > 1) spreads the simulation over the GPUs of the node — slabs of the domain,
 *     one-cell halo over RCCL/xGMI each neighbour pass — behind the SAME sphmi_upload / sphmi_advance / sphmi_download
 *     calls: the caller (one Julia process, src/SPHCellList.jl:883) does not change.
 */
#ifndef SPHMI_H
#define SPHMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPHMI_ABI_VERSION 5
#define SPHMI_MAX_DEVICES 16

/* status codes */
enum {
    SPHMI_OK              = 0,
    SPHMI_ERR_ARGUMENT    = 1,  /* bad config / null pointer / unsupported model tag          */
    SPHMI_ERR_DEVICE      = 2,  /* HIP runtime failure (text in sphmi_last_error)              */
    SPHMI_ERR_NUMERIC     = 3,  /* NaN / non-positive dt produced by the time-step criterion   */
    SPHMI_ERR_DOMAIN      = 4,  /* bounding cell grid exceeds the configured cell budget       */
    SPHMI_ERR_STATE       = 5   /* call sequence error (e.g. advance before upload)            */
};

/* model tags; values mirror the reference's dispatch types */
enum { SPHMI_KERNEL_WENDLAND_C2 = 0, SPHMI_KERNEL_CUBIC_SPLINE = 1 };   /* src/SPHKernels.jl:13-19,75-126 */
enum { SPHMI_KOUT_NONE = 0, SPHMI_KOUT_STORE = 1 };         /* KMode: src/SPHCellList.jl:90-116 */
enum { SPHMI_VISC_ZERO = 0, SPHMI_VISC_ARTIFICIAL = 1, SPHMI_VISC_LAMINAR = 2, SPHMI_VISC_LAMINAR_SPS = 3 };
                                                            /* src/SPHViscosityModels.jl:51-126  */
enum { SPHMI_DDT_NONE = 0, SPHMI_DDT_ZERO_GRAVITY_LINEAR = 1, SPHMI_DDT_LINEAR = 2, SPHMI_DDT_COMPLEX = 3 };
                                                            /* src/SPHDensityDiffusionModels.jl:30-188 */
enum { SPHMI_SHIFT_NONE = 0, SPHMI_SHIFT_PLANAR = 1 };      /* src/SPHCellList.jl:73-88,654-677 */
enum { SPHMI_MDBC_NONE = 0, SPHMI_MDBC_SIMPLE = 1 };        /* src/SimulationMetaDataConfiguration.jl:20-22 */
/* ParticleType values, src/SimulationGeometry.jl:10-14 */
enum { SPHMI_FLUID = 1, SPHMI_FIXED = 2, SPHMI_MOVING = 3 };

/*
 * Parameter block = the fields of SimulationConstants (src/SimulationConstantsConfiguration.jl:36-52)
 * and SPHKernelInstance (src/SPHKernels.jl:30-72) passed by value, plus the type parameters of
 * SimulationMetaData{D,T,SMode,KMode,BMode,LMode} (src/SimulationMetaDataConfiguration.jl:28-33)
 * and the model tag types as enums.
 */
typedef struct sphmi_config {
    int32_t struct_size;         /* = sizeof(sphmi_config); ABI guard                            */
    int32_t abi_version;         /* = SPHMI_ABI_VERSION                                          */
    int32_t dims;                /* Dimensions: 2 or 3                                           */
    int32_t host_float_bytes;    /* FloatType of the host arrays: 4 or 8                         */
    int32_t device_float_bytes;  /* arithmetic type of the kernels: 4, 8, or 0 = chosen by the library (sphmi_auto_device_float_bytes) */
    int32_t kernel;              /* SPHMI_KERNEL_*                                               */
    int32_t viscosity;           /* SPHMI_VISC_*                                                 */
    int32_t density_diffusion;   /* SPHMI_DDT_*                                                  */
    int32_t mdbc;                /* SPHMI_MDBC_*                                                 */
    int32_t device;              /* HIP device ordinal                                           */
    int32_t shifting;            /* SPHMI_SHIFT_* (SMode of SimulationMetaData)                  */
    int32_t kernel_output;       /* SPHMI_KOUT_* (KMode of SimulationMetaData)                   */
    int64_t n_particles;         /* length(SimParticles); below 2^27 (fp32 kernels) / 2^26 (fp64) per device: 32-bit gather offsets */
    int64_t max_cells;           /* cell budget of the dense bounding grid; 0 = default (1<<30: two 4-byte arrays with 25 % headroom, 10.7 GB at the limit, allocated on demand; SPHMI_ERR_DOMAIN when the device cannot hold them) */
    /* SimulationConstants */
    double rho0, dx, m0, alpha, g, c0, gamma, delta_phi, CFL, Cb, nu0;
    /* SPHKernelInstance */
    double k, h, h_inv, H, H_inv, H2, alphaD, eta2;
    /* SimulationConstants, continued (LaminarSPS) */
    double blin_constant, smagorinsky_constant;
    /* CubicSpline.eps (tensile correction, src/SPHKernels.jl:15-19,114-126) */
    double cubic_eps;
    /* Device list: n_devices <= 1 → one GPU, `device` above.  n_devices > 1 → the particle set is
     * split into n_devices slabs along one axis, slab r on HIP device devices[r]; halos and the per-step MAX-allreduce
     * travel over RCCL.  The same ordinal may appear more than once (several slabs share that GPU; transfers are then
     * stream-ordered device copies — the single-GPU test configuration).  slab_axis: 0 = chosen for balance, 1 / 2 / 3 =
     * x / y / z. */
    int32_t n_devices;
    int32_t slab_axis;
    int32_t devices[SPHMI_MAX_DEVICES];
} sphmi_config;

/* What the reference's SimulationLoop leaves in SimMetaData (src/SPHCellList.jl:679-685,:759). */
typedef struct sphmi_progress {
    int64_t iteration;        /* SimMetaData.Iteration                                            */
    int64_t steps_done;       /* steps executed by this call                                      */
    int64_t n_rebuilds;       /* UpdateNeighbors! executions since create                         */
    int64_t index_counter;    /* SimMetaData.IndexCounter = 1 + number of occupied cells          */
    double  total_time;       /* SimMetaData.TotalTime                                            */
    double  last_dt;          /* SimMetaData.CurrentTimeStep                                      */
    double  delta_x;          /* the loop-local Δx accumulator (src/SPHCellList.jl:739,744,760)   */
} sphmi_progress;

typedef struct sphmi_handle sphmi_handle;

/* Library identification: "sphmi <abi> hip gfx950 ..." — static string. */
const char* sphmi_backend_info(void);

/* Text of the last error on this handle (or of the last failed sphmi_create when h == NULL). */
const char* sphmi_last_error(const sphmi_handle* h);

/* Allocate device state for cfg->n_particles particles.  Replaces the allocations at
 * src/SPHCellList.jl:825,837,840-844 (support arrays, per-thread copies, ParticleRanges,
 * UniqueCells, CellDict, sort scratch). */
int sphmi_create(const sphmi_config* cfg, sphmi_handle** out);

/* The arithmetic `device_float_bytes = 0` resolves to for this configuration: 4 (fp32 kernels, double-float state) when every term of
 * the path is continuous in the positions — the kernel support ends where the kernel and its gradient vanish (H >= 2h, the default
 * k = 2 of src/SPHKernels.jl:57-60, so nothing jumps at the r^2 <= H^2 cut of src/SPHCellList.jl:275) and there is no mDBC; 8 (fp64
 * kernels) when the kernel is cut off before it vanishes (k < 2: example/DucklingMDBC.jl, example/MovingSquare2d.jl) or mDBC is on
 * (src/SPHCellList.jl:598-622: "no neighbour -> keep", the Shepard fallback and the |det A| >= 1e-3 switch are discontinuities) — there
 * an fp32 trajectory takes the other branch a step early or late and leaves the reference's state by more than 1e-5 on single
 * particles.  ("H >= 2h" is tested with a relative slack of 1e-12: H = k*h is formed in floating point on the caller's side.)  A handle too
 * large for the fp64 kernels (more than 2^26 - 1 particles per device) stays with fp32 whatever the policy says.
 * sphmi_device_float_bytes: what a handle runs. */
int32_t sphmi_auto_device_float_bytes(const sphmi_config* cfg);
int sphmi_device_float_bytes(const sphmi_handle* h, int32_t* device_float_bytes_out);
int sphmi_destroy(sphmi_handle* h);

/*
 * Copy the SimParticles fields the hot path reads (src/PreProcess.jl:114) to the device.
 * position / velocity / acceleration / ghost_points: N*D host floats (AoS); density: N host floats;
 * type: N ParticleType bytes; id: N Int64; group_marker: N UInt64 (may be NULL);
 * ghost_points may be NULL when mdbc == SPHMI_MDBC_NONE.  GravityFactor / MotionLimiter are
 * derived from `type` exactly as src/PreProcess.jl:78-98 does.
 * Also resets Positionₙ⁺ to zero as AllocateSupportDataStructures does (src/PreProcess.jl:131).
 */
int sphmi_upload(sphmi_handle* h,
                 const void* position, const void* velocity, const void* acceleration,
                 const void* density, const uint8_t* type, const int64_t* id,
                 const uint64_t* group_marker, const void* ghost_points);

/*
 * Pre-processing on the device for the case the headline is quoted on.  The reference builds
 * SimParticles from CSV point lists (LoadSpecificCSV / AllocateDataStructures, src/PreProcess.jl:45-119) and ships the
 * 3-D dam break only at Dp 0.02; sphmi_generate_dam_break_3d fills the handle with the lattice of that case at any
 * spacing dp — same node set, order, IDs (1 … N, boundary first), Type / GroupMarker and hydrostatic densities as the
 * files (and as the host generator the tests compare with) — without host arrays or an upload, and leaves the handle
 * in the state sphmi_upload leaves it in.  cfg->n_particles must equal n_bound + n_fluid of sphmi_dam_break_3d_count;
 * rho0, g, c0 come from the handle's config.  3-D single-device handles.
 */
int sphmi_dam_break_3d_count(double dp, int64_t* n_bound_out, int64_t* n_fluid_out);
int sphmi_generate_dam_break_3d(sphmi_handle* h, double dp);

/* Set / read SimMetaData.Iteration and SimMetaData.TotalTime (they live in the host struct). */
int sphmi_set_clock(sphmi_handle* h, int64_t iteration, double total_time);

/*
 * Output without stalling the run: sphmi_download_begin snapshots the requested fields on the
 * device in stream order and starts the device→host copies on a second stream; the caller may call sphmi_advance
 * right away and must not touch the host arrays until sphmi_download_end returns.  sphmi_download = begin + end.
 * sphmi_host_register page-locks a host array the caller will hand in again and again (the fields of the
 * StructArray live for the whole run): registered arrays are written by the copy engine WHILE the caller advances;
 * arrays that are not registered are filled from the device-side snapshot inside sphmi_download_end, through a
 * page-locked bounce buffer of the handle (the library never hands a pageable pointer to the runtime: its cached pins
 * of the caller's pages go stale when the allocator re-uses the addresses).  The contract is the same either way: the
 * arrays hold the snapshot when sphmi_download_end returns.  The caller must unregister (or destroy the handle) before
 * freeing a registered array.
 */
int sphmi_download_begin(sphmi_handle* h,
                         void* position, void* velocity, void* acceleration, void* density, void* pressure,
                         int64_t* id, uint8_t* type, uint64_t* group_marker, void* ghost_points, int64_t* cells);
int sphmi_download_end(sphmi_handle* h);
/* Components per vector of the downloaded Position / Velocity / Acceleration / GhostPoints: `dims` (default, the
 * SVector{D} layout of SimParticles) or 3 — the point layout of the VTKHDF writer, which pads 2-D vectors with a zero
 * (to_3d!, src/ProduceHDFVTK.jl:251-325): the arrays handed to sphmi_download* then hold n×3 values and can be appended
 * to the `Points` / PointData datasets as they are.  Cells stay n×dims. */
int sphmi_set_output_components(sphmi_handle* h, int components);
/* (multi-device handles stage through page-locked buffers of their own and merge the slabs on the host: registering the
 * caller's arrays is accepted and gains nothing there) */
int sphmi_host_register(sphmi_handle* h, void* ptr, int64_t bytes);
int sphmi_host_unregister(sphmi_handle* h, void* ptr);

/*
 * StoreKernelOutput (src/SPHCellList.jl:106-116): Kernel[i] = Σⱼ Wᵢⱼ and KernelGradient[i] = Σⱼ ∇ᵢWᵢⱼ of the last
 * neighbour pass, host float type, n and n×dims values, current (cell-sorted) order.  Needs kernel_output = STORE.
 */
int sphmi_download_kernel_output(sphmi_handle* h, void* kernel, void* kernel_gradient);

/*
 * The sort as a permutation.  The reference's sort! permutes ALL 17 fields of the SimParticles StructArray
 * (src/SPHCellList.jl:142); the engine carries the ten the hot path touches and sphmi_download returns those in the
 * current cell-sorted order.  prev_row (N Int64, 0-based) says where every row came from: row i of the arrays
 * sphmi_download delivers NOW was row prev_row[i] at the previous call of sphmi_download_permutation (at sphmi_upload
 * for the first call), so the caller brings the fields the engine does not carry (GhostNormals, ChunkID, user columns)
 * along with ONE gather per field — no sort on the host.  The engine has the permutation from its own sort (a 4-byte
 * column that travels with the particles); multi-device handles derive it from the ID column.  Not for rank-mode handles.
 */
int sphmi_download_permutation(sphmi_handle* h, int64_t* prev_row);

/*
 * MotionDetails of the Geometry with this GroupMarker (src/SimulationGeometry.jl:17-22): particles of Type Moving
 * in that group get Velocity = velocity·direction while start_time <= TotalTime <= start_time + duration (0
 * otherwise) and are displaced by Velocity·dt/2 before each neighbour pass — ProgressMotion,
 * src/SPHCellList.jl:575-596, called at :765 and :787.  `direction` holds `dims` doubles.  At most 16 groups.
 */
int sphmi_set_motion(sphmi_handle* h, uint64_t group_marker, double velocity, double start_time, double duration,
                     const double* direction);

/*
 * One SimulationLoop call (src/SPHCellList.jl:727-805): reset Δx = 1 + h, then step
 * `while TotalTime <= t_target`.  max_steps < 0 means unbounded; otherwise the loop also stops
 * after max_steps steps (used by tests and by the benchmark's fixed-step window).
 */
int sphmi_advance(sphmi_handle* h, double t_target, int64_t max_steps, sphmi_progress* out);

/*
 * Copy particle state back in the engine's current (cell-sorted) order — the same order the
 * reference leaves SimParticles in, because its sort! permutes every field (src/SPHCellList.jl:142).
 * Any pointer may be NULL to skip that field.  cells: N*D Int64 (Particles.Cells).
 */
int sphmi_download(sphmi_handle* h,
                   void* position, void* velocity, void* acceleration,
                   void* density, void* pressure,
                   int64_t* id, uint8_t* type, uint64_t* group_marker,
                   void* ghost_points, int64_t* cells);

/*
 * Parity hook: rebuild the cell list for the current positions, run Pressure! and ONE
 * NeighborLoop!+ReductionStep! (src/SPHCellList.jl:771,774-775) on the current state and return
 * dρdtI (N) and Acceleration (N*D, no gravity) in cell-sorted order; also sorts the particles
 * (as UpdateNeighbors! does) but does not advance time.  apply_mdbc != 0 additionally runs
 * ApplyMDBCBeforeHalf! (:772) between Pressure! and the pair loop.  Multi-device handles: the rebuild is the collective
 * one (migration, fresh ghost layers) and the rows come back merged, in the order one device would hold.
 */
int sphmi_forces_once(sphmi_handle* h, int apply_mdbc, void* drhodt, void* acceleration);

/* Occupied cells in sort order: cells_out receives (index_counter-1)*D Int64 values
 * (UniqueCells[2:IndexCounter], src/SPHCellList.jl:148-157); n_out the count. */
int sphmi_unique_cells(sphmi_handle* h, int64_t* cells_out, int64_t capacity, int64_t* n_out);

/*
 * Per-phase device seconds accumulated since create, under the reference's TimerOutputs labels
 * (src/SPHCellList.jl:748-800).  names_out receives pointers to static strings.  The step phases are timed with
 * HIP events on every 8th step and scaled by 8 (an event pair per phase and step costs more than a small pass);
 * rebuilds are always timed.
 */
int sphmi_timers(sphmi_handle* h, int32_t capacity, const char** names_out, double* seconds_out,
                 int64_t* calls_out, int32_t* n_out);

/* Raw device pointers of the packed neighbour stream (state set A) and the live particle count.  The two packets of a
 * particle are interleaved in ONE array of records: packet h of particle i is pk_h[2*i] (pk1 == pk0 + 1 packet). */
int sphmi_device_ptrs(sphmi_handle* h, void** pk0, void** pk1, int64_t* n_local);

/* ---- measurement hooks (bench.py) ------------------------------------------------------- */
/* Average duration in ms (HIP events on the engine's stream) of the neighbour+force kernel over the
 * launches since the last reset (sampled: the launches of every 8th step), and the number of launches. */
int sphmi_force_kernel_stats(sphmi_handle* h, int reset, double* avg_ms_out, int64_t* launches_out);

/* ---- multi-GPU handles ------------------------------------------------------------------------------------
 * sphmi_create with cfg->n_devices > 1: all slabs in THIS process (what the reference's single Julia process needs).
 * sphmi_create_rank: the same slab driver with ONE slab per process and the other slabs behind RCCL — rank r of `world`
 * processes (torchrun: one process per GPU).  Every process passes the SAME full particle set to sphmi_upload and keeps
 * its slab; sphmi_download then returns the particles this process owns (sphmi_owned_count of them, at most
 * cfg->n_particles).  `unique_id`: the 128 bytes sphmi_rccl_unique_id produced on rank 0, distributed by the launcher
 * (a file, MPI, torch.distributed's store …).  cfg->device is this rank's GPU.
 * SPHMI_TRANSPORT=shm in the environment puts the peers behind a POSIX shared-memory segment of the node instead of
 * RCCL (messages and reductions staged through the host, csrc/sphmi_shm.h): RCCL refuses two ranks on one device, so
 * this is how the rank-mode driver runs — and is tested — with more ranks than GPUs.  Every wait has a deadline
 * (SPHMI_SHM_TIMEOUT seconds, default 120). */
int sphmi_rccl_unique_id(void* id_out /* 128 bytes */);
/* Binds RCCL in this process (dlopen + every symbol the slab driver calls) WITHOUT asking it for an id: ncclGetUniqueId starts a
 * bootstrap root — a listening socket and a thread — per call, so only the rank that hands the id out should call it; the other ranks
 * check with this that their process could join.  SPHMI_ERR_DEVICE + sphmi_last_error(NULL) when RCCL is out of reach. */
int sphmi_rccl_probe(void);
int sphmi_create_rank(const sphmi_config* cfg, int32_t rank, int32_t world, const void* unique_id, sphmi_handle** out);
int sphmi_owned_count(sphmi_handle* h, int64_t* n_out);   /* particles sphmi_download returns (any handle)        */
typedef struct sphmi_multi_info {
    int32_t world, n_local;        /* slabs in total / held by this handle                                         */
    int32_t axis, halo_width;      /* slab axis (0 = x …); ghost-layer width in cell columns (1; 2 + off with mDBC) */
    int32_t transport;             /* 0 = stream-ordered device copies, 1 = RCCL, 2 = host shared memory (below)    */
    int32_t reserved;              /* how the four per-step maxima travel: 0 = the transport's collective (ncclAllReduce), 1 = device mailboxes ($SPHMI_EXCHANGE=mailbox) */
    int64_t n_recuts;              /* rebuilds at which the cuts moved (load balance by work)                        */
    int64_t cuts[SPHMI_MAX_DEVICES];    /* cuts[r-1] = first cell column of slab r                                  */
    int64_t n_live[SPHMI_MAX_DEVICES];  /* particles incl. ghost copies currently held per local slab               */
} sphmi_multi_info;
int sphmi_multi_info_get(sphmi_handle* h, sphmi_multi_info* out);
/* (Test hooks of the slab driver — the shared-memory transport's self-test, initial cuts, host-only planning, the work
 * measure of the re-cut — are declared in include/sphmi_internal.h; they are not part of the drop-in boundary.) */

#ifdef __cplusplus
}
#endif
#endif /* SPHMI_H */
