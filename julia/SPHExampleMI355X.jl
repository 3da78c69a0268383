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
# ZeroGravityLinear / Linear / Complex density diffusion and leaves user-defined SPHViscosity / SPHDensityDiffusion
# subtypes (example/Dambreak2dMDBC.jl:46-66) on the CPU path — no method is overwritten.
#
# Environment: SPHMI_LIB (path of libsphmi.so), SPHMI_DEVICE_FLOAT_BYTES (4 = fp32 kernels, default; 8 = fp64),
# SPHMI_DEVICES ("0" default; "0,1,2,3,4,5,6,7" = one slab per GPU, halos over RCCL — same calls, see sphmi.h).
#
# EXPERIMENTAL: the build image has no Julia, so this file has never been executed.  struct layout and ABI version
# are asserted against the library at first use (sphmi_create refuses a mismatching struct_size / abi_version).
module SPHExampleMI355X

using SPHExample, StaticArrays
import SPHExample.SPHCellList: SimulationLoop, next_output_time

const LIB = get(ENV, "SPHMI_LIB", "libsphmi.so")
const ABI_VERSION = Int32(3)

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
const BuiltinDDT = Union{ZeroGravityLinearDensityDiffusion,LinearDensityDiffusion,ComplexDensityDiffusion}
tag(::ZeroViscosity) = Int32(0); tag(::ArtificialViscosity) = Int32(1); tag(::Laminar) = Int32(2); tag(::LaminarSPS) = Int32(3)
tag(::ZeroGravityLinearDensityDiffusion) = Int32(1); tag(::LinearDensityDiffusion) = Int32(2); tag(::ComplexDensityDiffusion) = Int32(3)

const HANDLES = IdDict{Any,Ptr{Cvoid}}()          # SimParticles (identity) → engine handle
atexit(() -> foreach(h -> ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), h), values(HANDLES)))

function check(h, rc)
    rc == 0 || error("libsphmi status $rc: " * unsafe_string(ccall((:sphmi_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))
end

function open_handle(SimDensityDiffusion, SimViscosity, SimKernel, SimMetaData::SimulationMetaData{D,T,S,K,B,L}, SimConstants, P, MotionDefinition) where {D,T,S,K,B,L}
    devs = parse.(Int32, split(get(ENV, "SPHMI_DEVICES", "0"), ","))
    cfg = SphmiConfig(sizeof(SphmiConfig), ABI_VERSION, D, sizeof(T), parse(Int32, get(ENV, "SPHMI_DEVICE_FLOAT_BYTES", "4")),
                      SimKernel.kernel isa CubicSpline ? 1 : 0, tag(SimViscosity), tag(SimDensityDiffusion), B <: SimpleMDBC ? 1 : 0,
                      devs[1], S <: PlanarShifting ? 1 : 0, K <: StoreKernelOutput ? 1 : 0, length(P), 0,
                      SimConstants.ρ₀, SimConstants.dx, SimConstants.m₀, SimConstants.α, SimConstants.g, SimConstants.c₀,
                      SimConstants.γ, SimConstants.δᵩ, SimConstants.CFL, SimConstants.Cb, SimConstants.ν₀,
                      SimKernel.k, SimKernel.h, SimKernel.h⁻¹, SimKernel.H, SimKernel.H⁻¹, SimKernel.H², SimKernel.αD, SimKernel.η²,
                      SimConstants.BlinConstant, SimConstants.SmagorinskyConstant,
                      SimKernel.kernel isa CubicSpline ? Float64(SimKernel.kernel.eps) : 0.0,
                      length(devs), 0, ntuple(i -> i <= length(devs) ? devs[i] : Int32(0), 16))
    href = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:sphmi_create, LIB), Cint, (Ref{SphmiConfig}, Ref{Ptr{Cvoid}}), cfg, href))
    h = href[]
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
    return h
end

# One output interval on the device: the contract of src/SPHCellList.jl:727-805 — advance until TotalTime > next output
# time, leave the state in SimParticles (cell-sorted, every field permuted alike) and the counters in SimMetaData.
function SimulationLoop(SimDensityDiffusion::BuiltinDDT, SimViscosity::BuiltinViscosity, SimKernel,
                        SimMetaData::SimulationMetaData{D,T,S,K,B,L}, SimConstants, SimParticles, Stencil, ParticleRanges,
                        UniqueCells, CellDict, SortingScratchSpace, SimThreadedArrays, dρdtI, Velocityₙ⁺, Positionₙ⁺, ρₙ⁺,
                        ∇Cᵢ, ∇◌rᵢ, MotionDefinition) where {D,T,S,K,B,L}
    P = SimParticles
    N = length(P)
    h = get!(() -> open_handle(SimDensityDiffusion, SimViscosity, SimKernel, SimMetaData, SimConstants, P, MotionDefinition), HANDLES, P)
    prog = SphmiProgress()
    check(h, ccall((:sphmi_advance, LIB), Cint, (Ptr{Cvoid}, Float64, Int64, Ref{SphmiProgress}), h, Float64(next_output_time(SimMetaData)), -1, prog))
    SimMetaData.Iteration, SimMetaData.CurrentTimeStep, SimMetaData.TotalTime = prog.iteration, T(prog.last_dt), T(prog.total_time)
    old_id = copy(P.ID)
    typ = Vector{UInt8}(undef, N); cells = Vector{Int64}(undef, N * D)
    GC.@preserve P typ cells begin
        check(h, ccall((:sphmi_download, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{UInt64}, Ptr{Cvoid}, Ptr{Int64}),
            h, pointer(P.Position), pointer(P.Velocity), pointer(P.Acceleration), pointer(P.Density), pointer(P.Pressure),
            pointer(P.ID), pointer(typ), pointer(P.GroupMarker), B <: SimpleMDBC ? pointer(P.GhostPoints) : C_NULL, pointer(cells)))
        K <: StoreKernelOutput && check(h, ccall((:sphmi_download_kernel_output, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), h, pointer(P.Kernel), pointer(P.KernelGradient)))
    end
    # fields the engine does not carry follow the particles through the sort by ID (the reference's sort! permutes all 17)
    perm = Vector{Int}(undef, N); perm[sortperm(P.ID)] = sortperm(old_id)             # new row i was old row perm[i]
    P.GhostNormals .= P.GhostNormals[perm]; P.ChunkID .= P.ChunkID[perm]
    K <: StoreKernelOutput || (P.Kernel .= P.Kernel[perm]; P.KernelGradient .= P.KernelGradient[perm])
    @inbounds for i in 1:N
        P.Type[i] = ParticleType(typ[i])
        P.Cells[i] = CartesianIndex(ntuple(d -> Int(cells[(i - 1) * D + d]), D))
        P.GravityFactor[i] = typ[i] == 1 ? -one(T) : (typ[i] == 3 ? one(T) : zero(T))   # src/PreProcess.jl:78-100
        P.MotionLimiter[i] = typ[i] == 1 ? one(T) : zero(T)
        P.BoundaryBool[i] = typ[i] == 1 ? 0x00 : 0x01
    end
    SimMetaData.IndexCounter = prog.index_counter
    if SimMetaData.ExportGridCells     # UniqueCells[2:IndexCounter] for save_grid (:890-893); slot 1 is the reference's dummy entry (:145-147)
        nref = Ref{Int64}(0)
        ucells = Vector{Int64}(undef, N * D)
        GC.@preserve ucells check(h, ccall((:sphmi_unique_cells, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Int64}), h, pointer(ucells), N, nref))
        @inbounds for k in 1:min(Int(nref[]), length(UniqueCells) - 1)
            UniqueCells[k + 1] = CartesianIndex(ntuple(d -> Int(ucells[(k - 1) * D + d]), D))
        end
    end
    if SimMetaData.TotalTime > SimMetaData.SimulationTime                              # last interval (:909): release the GPUs
        ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), h); delete!(HANDLES, P)
    end
    return nothing
end

end # module
