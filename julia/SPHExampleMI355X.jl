# SPHExampleMI355X.jl — runs SPHExample's hot loop on MI355X GPUs through libsphmi.so (C ABI: include/sphmi.h).
#
#     using SPHExample
#     include("path/to/julia/SPHExampleMI355X.jl")         # before the RunSimulation(...) call; nothing else changes
#
# It adds ONE method: SPHExample.SPHCellList.SimulationLoop for the built-in model tags.  That function is what
# RunSimulation calls once per output interval (src/SPHCellList.jl:883; generic definition :727-733), so example/*.jl,
# RunSimulation itself, logging, ProgressMeter, TimerOutputs, save_particles / save_grid and the VTKHDF writer stay the
# reference's own code.  The method is MORE SPECIFIC than the reference's (concrete Union of tag types in the first
# two arguments): dispatch picks it for ZeroViscosity / ArtificialViscosity / Laminar / LaminarSPS with
# Zero / ZeroGravityLinear / Linear / Complex density diffusion and leaves user-defined SPHViscosity / SPHDensityDiffusion
# subtypes (example/Dambreak2dMDBC.jl:46-66) on the CPU path — no method is overwritten.
#
# What one call costs on the host (round 3): the device → host copies of the fields the engine carries
# (sphmi_download_begin … _end, straight into the columns of the StructArray, which are page-locked once), ONE gather per
# passive column with the permutation the engine hands over (sphmi_download_permutation: the reference's sort! permutes all
# 17 columns, src/SPHCellList.jl:142; the engine carries ten) — no sortperm, no per-particle loop: Type, GravityFactor,
# MotionLimiter and BoundaryBool are per-particle constants and follow the same gather, Cells are written in place
# (a CartesianIndex{D} is D Int64s).  The gathers run while the copies are in flight.
#
# Environment: SPHMI_LIB (path of libsphmi.so), SPHMI_DEVICE_FLOAT_BYTES (0, default = the library chooses — fp32 kernels when every
# term of the path is continuous: the kernel vanishes at its cut-off, SimKernel.k >= 2, and BMode is NoMDBC (Dambreak3d.jl); fp64
# kernels for DucklingMDBC.jl / MovingSquare2d.jl (k < 2) and every SimpleMDBC run (sphmi_auto_device_float_bytes, include/sphmi.h);
# 4 = fp32; 8 = fp64),
# SPHMI_DEVICES ("0" default; "0,1,2,3,4,5,6,7" = one slab per GPU, halos over RCCL — same calls, see sphmi.h).
#
# EXPERIMENTAL: the build image has no Julia, so this file has never been executed.  struct layout and ABI version
# are asserted against the library at first use (sphmi_create refuses a mismatching struct_size / abi_version).
module SPHExampleMI355X

using SPHExample, StaticArrays, TimerOutputs
import SPHExample.SPHCellList: SimulationLoop, next_output_time

const LIB = get(ENV, "SPHMI_LIB", "libsphmi.so")
const ABI_VERSION = Int32(5)

struct SphmiConfig                       # struct sphmi_config, field for field (include/sphmi.h)
    struct_size::Int32; abi_version::Int32; dims::Int32; host_float_bytes::Int32; device_float_bytes::Int32
    kernel::Int32; viscosity::Int32; density_diffusion::Int32; mdbc::Int32; device::Int32
    shifting::Int32; kernel_output::Int32
    n_particles::Int64; max_cells::Int64
    rho0::Float64; dx::Float64; m0::Float64; alpha::Float64; g::Float64; c0::Float64; gamma::Float64
    delta_phi::Float64; CFL::Float64; Cb::Float64; nu0::Float64
    k::Float64; h::Float64; h_inv::Float64; H::Float64; H_inv::Float64; H2::Float64; alphaD::Float64; eta2::Float64
    blin_constant::Float64; smagorinsky_constant::Float64; cubic_eps::Float64
    n_devices::Int32; slab_axis::Int32; devices::NTuple{16,Int32}
end
mutable struct SphmiProgress
    iteration::Int64; steps_done::Int64; n_rebuilds::Int64; index_counter::Int64
    total_time::Float64; last_dt::Float64; delta_x::Float64
    SphmiProgress() = new(0, 0, 0, 0, 0.0, 0.0, 0.0)
end

const BuiltinViscosity = Union{ZeroViscosity,ArtificialViscosity,Laminar,LaminarSPS}
const BuiltinDDT = Union{ZeroDensityDiffusion,ZeroGravityLinearDensityDiffusion,LinearDensityDiffusion,ComplexDensityDiffusion}
tag(::ZeroViscosity) = Int32(0); tag(::ArtificialViscosity) = Int32(1); tag(::Laminar) = Int32(2); tag(::LaminarSPS) = Int32(3)
tag(::ZeroDensityDiffusion) = Int32(0); tag(::ZeroGravityLinearDensityDiffusion) = Int32(1); tag(::LinearDensityDiffusion) = Int32(2); tag(::ComplexDensityDiffusion) = Int32(3)

# per simulation: the engine handle and the host scratch that lives as long as it (page-locked once, reused every interval)
mutable struct Session
    h::Ptr{Cvoid}
    prev_row::Vector{Int64}              # sphmi_download_permutation: row i now was row prev_row[i] (0-based) at the previous call
    secs::Vector{Float64}                # sphmi_timers at the previous call (device seconds per phase), calls likewise
    calls::Vector{Int64}
    perm::Vector{Int}                    # the same, 1-based
    ucells::Vector{Int64}
end
const SESSIONS = IdDict{Any,Session}()            # SimParticles (identity) → session
atexit(() -> foreach(s -> ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), s.h), values(SESSIONS)))

function check(h, rc)
    rc == 0 || error("libsphmi status $rc: " * unsafe_string(ccall((:sphmi_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))
end

# the columns sphmi_download writes every interval: page-locked once (the arrays of the StructArray live for the whole run;
# a multi-device handle accepts the call and stages through its own buffers).  Pinning is an optimisation only — an array that
# cannot be page-locked (locked-memory limit, a range the runtime already knows) is filled through the engine's bounce buffer
# in sphmi_download_end — so a refusal is not an error.
pin(h, a::Array) = isempty(a) || ccall((:sphmi_host_register, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h, pointer(a), sizeof(a))

function open_session(SimDensityDiffusion, SimViscosity, SimKernel, SimMetaData::SimulationMetaData{D,T,S,K,B,L}, SimConstants, P, MotionDefinition) where {D,T,S,K,B,L}
    devs = parse.(Int32, split(get(ENV, "SPHMI_DEVICES", "0"), ","))
    N = length(P)
    cfg = SphmiConfig(sizeof(SphmiConfig), ABI_VERSION, D, sizeof(T), parse(Int32, get(ENV, "SPHMI_DEVICE_FLOAT_BYTES", "0")),
                      SimKernel.kernel isa CubicSpline ? 1 : 0, tag(SimViscosity), tag(SimDensityDiffusion), B <: SimpleMDBC ? 1 : 0,
                      devs[1], S <: PlanarShifting ? 1 : 0, K <: StoreKernelOutput ? 1 : 0, N, 0,
                      SimConstants.ρ₀, SimConstants.dx, SimConstants.m₀, SimConstants.α, SimConstants.g, SimConstants.c₀,
                      SimConstants.γ, SimConstants.δᵩ, SimConstants.CFL, SimConstants.Cb, SimConstants.ν₀,
                      SimKernel.k, SimKernel.h, SimKernel.h⁻¹, SimKernel.H, SimKernel.H⁻¹, SimKernel.H², SimKernel.αD, SimKernel.η²,
                      SimConstants.BlinConstant, SimConstants.SmagorinskyConstant,
                      SimKernel.kernel isa CubicSpline ? Float64(SimKernel.kernel.eps) : 0.0,
                      length(devs), 0, ntuple(i -> i <= length(devs) ? devs[i] : Int32(0), 16))
    href = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:sphmi_create, LIB), Cint, (Ref{SphmiConfig}, Ref{Ptr{Cvoid}}), cfg, href))
    h = href[]
    try                                   # (from here on the handle exists and is not yet in SESSIONS: an error must not leak it)
    for (group, m) in enumerate(MotionDefinition)                                  # RunSimulation's table, :846-850
        m === nothing && continue
        dir = Float64[m.Direction...]
        GC.@preserve dir check(h, ccall((:sphmi_set_motion, LIB), Cint, (Ptr{Cvoid}, UInt64, Float64, Float64, Float64, Ptr{Float64}),
                                        h, UInt64(group), Float64(m.Velocity), Float64(m.StartTime), Float64(m.Duration), pointer(dir)))
    end
    typ = Vector{UInt8}(UInt8.(P.Type))
    GC.@preserve P typ check(h, ccall((:sphmi_upload, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int64}, Ptr{UInt64}, Ptr{Cvoid}),
        h, pointer(P.Position), pointer(P.Velocity), pointer(P.Acceleration), pointer(P.Density), pointer(typ), pointer(P.ID),
        pointer(P.GroupMarker), B <: SimpleMDBC ? pointer(P.GhostPoints) : C_NULL))
    check(h, ccall((:sphmi_set_clock, LIB), Cint, (Ptr{Cvoid}, Int64, Float64), h, SimMetaData.Iteration, SimMetaData.TotalTime))
    for a in (P.Position, P.Velocity, P.Acceleration, P.Density, P.Pressure, P.ID, P.GroupMarker, P.Cells)
        pin(h, a)
    end
    B <: SimpleMDBC && pin(h, P.GhostPoints)
    return Session(h, Vector{Int64}(undef, N), zeros(8), zeros(Int64, 8), Vector{Int}(undef, N), Vector{Int64}(undef, SimMetaData.ExportGridCells ? N * D : 0))
    catch
        ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), h)
        rethrow()
    end
end

# The engine's phases under the reference's own TimerOutputs labels (src/SPHCellList.jl:748-798: "01 Update TimeStep", "02a Actual
# Calculate IndexCounter", "04 Apply MDBC before Half TimeStep", "05 First NeighborLoop", "08 Second NeighborLoop"), nested under
# "00 SimulationLoop" (:883) like the reference's: a section is opened empty, then credited with the DEVICE seconds and calls the
# engine measured since the previous interval (`accumulated_data` is TimerOutputs' own field: ncalls, time in ns).
function forward_timers!(to::TimerOutput, s::Session)
    names = Vector{Cstring}(undef, 8); secs = zeros(8); calls = zeros(Int64, 8); n = Ref{Int32}(0)
    check(s.h, ccall((:sphmi_timers, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Cstring}, Ptr{Float64}, Ptr{Int64}, Ref{Int32}), s.h, 8, names, secs, calls, n))
    for k in 1:min(Int(n[]), 5)
        label = unsafe_string(names[k])
        @timeit to label nothing
        d = (isempty(to.timer_stack) ? to : to.timer_stack[end])[label].accumulated_data
        d.time += round(Int64, (secs[k] - s.secs[k]) * 1e9); d.ncalls += calls[k] - s.calls[k] - 1
    end
    s.secs .= secs; s.calls .= calls
end

# the reference's sort! permutes every column (src/SPHCellList.jl:142): the ones the engine does not carry follow by gather
permute_column!(a::AbstractVector, perm) = (a .= a[perm]; nothing)

# One output interval on the device: the contract of src/SPHCellList.jl:727-805 — advance until TotalTime > next output
# time, leave the state in SimParticles (cell-sorted, every field permuted alike) and the counters in SimMetaData.
function SimulationLoop(SimDensityDiffusion::BuiltinDDT, SimViscosity::BuiltinViscosity, SimKernel,
                        SimMetaData::SimulationMetaData{D,T,S,K,B,L}, SimConstants, SimParticles, Stencil, ParticleRanges,
                        UniqueCells, CellDict, SortingScratchSpace, SimThreadedArrays, dρdtI, Velocityₙ⁺, Positionₙ⁺, ρₙ⁺,
                        ∇Cᵢ, ∇◌rᵢ, MotionDefinition) where {D,T,S,K,B,L}
    P = SimParticles
    s = get!(() -> open_session(SimDensityDiffusion, SimViscosity, SimKernel, SimMetaData, SimConstants, P, MotionDefinition), SESSIONS, P)
    h = s.h
    prog = SphmiProgress()
    check(h, ccall((:sphmi_advance, LIB), Cint, (Ptr{Cvoid}, Float64, Int64, Ref{SphmiProgress}), h, Float64(next_output_time(SimMetaData)), -1, prog))
    SimMetaData.Iteration, SimMetaData.CurrentTimeStep, SimMetaData.TotalTime = prog.iteration, T(prog.last_dt), T(prog.total_time)
    SimMetaData.IndexCounter = prog.index_counter
    forward_timers!(SimMetaData.HourGlass, s)
    GC.@preserve P s begin
        # the carried fields: snapshot on the device, copies on a second stream, straight into the StructArray's columns
        # (Cells: a Vector{CartesianIndex{D}} is N·D Int64; Type is a per-particle constant and follows the gather below)
        check(h, ccall((:sphmi_download_begin, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{UInt64}, Ptr{Cvoid}, Ptr{Int64}),
            h, pointer(P.Position), pointer(P.Velocity), pointer(P.Acceleration), pointer(P.Density), pointer(P.Pressure),
            pointer(P.ID), C_NULL, pointer(P.GroupMarker), B <: SimpleMDBC ? pointer(P.GhostPoints) : C_NULL,
            Ptr{Int64}(pointer(P.Cells))))
        # while they are in flight: the sort as a permutation, and one gather per passive column
        check(h, ccall((:sphmi_download_permutation, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), h, pointer(s.prev_row)))
        s.perm .= s.prev_row .+ 1
        for col in (P.Type, P.GravityFactor, P.MotionLimiter, P.BoundaryBool, P.GhostNormals, P.ChunkID)
            permute_column!(col, s.perm)
        end
        if K <: StoreKernelOutput
            check(h, ccall((:sphmi_download_kernel_output, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, pointer(P.Kernel), pointer(P.KernelGradient)))
        else
            permute_column!(P.Kernel, s.perm); permute_column!(P.KernelGradient, s.perm)
        end
        if SimMetaData.ExportGridCells     # UniqueCells[2:IndexCounter] for save_grid (:890-893); slot 1 is the reference's dummy entry (:145-147)
            nref = Ref{Int64}(0)
            check(h, ccall((:sphmi_unique_cells, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Int64}), h, pointer(s.ucells), length(P), nref))
            @inbounds for k in 1:min(Int(nref[]), length(UniqueCells) - 1)
                UniqueCells[k + 1] = CartesianIndex(ntuple(d -> Int(s.ucells[(k - 1) * D + d]), D))
            end
        end
        check(h, ccall((:sphmi_download_end, LIB), Cint, (Ptr{Cvoid},), h))
    end
    if SimMetaData.TotalTime > SimMetaData.SimulationTime                              # last interval (:909): release the GPUs
        ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), h); delete!(SESSIONS, P)
    end
    return nothing
end

end # module
