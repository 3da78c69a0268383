# SPHExampleMI355X.jl — thin ccall shim that lets the unchanged SPHExample driver scripts run their hot
# path on an MI355X through libsphmi.so (C ABI: include/sphmi.h).
#
# WRITTEN BLIND: there is no Julia toolchain in the build image, so this file has never been executed.
# It is deliberately small: every line is either a field copy into the C parameter block or a ccall.
#
# Usage (in an example script, e.g. example/Dambreak3d.jl):
#     using SPHExample
#     include("path/to/julia/SPHExampleMI355X.jl"); using .SPHExampleMI355X
#     ENV["SPHMI_LIB"] = "/path/to/sphexample_amd/libsphmi.so"
#     RunSimulationMI355X(SimGeometry=..., SimMetaData=..., SimConstants=..., SimKernel=..., SimLogger=...,
#                         SimParticles=..., SimViscosity=..., SimDensityDiffusion=...)
# i.e. the keyword signature of SPHExample.RunSimulation (src/SPHCellList.jl:808-817).  Everything outside the
# call to SimulationLoop (:883) — VTKHDF output, logging, progress meter — stays the reference's own code.
module SPHExampleMI355X

export RunSimulationMI355X

using SPHExample
using StaticArrays
import StructArrays: StructArray

const LIB = get(ENV, "SPHMI_LIB", "libsphmi.so")

# struct sphmi_config, field for field (include/sphmi.h)
struct SphmiConfig
    struct_size::Int32; abi_version::Int32; dims::Int32; host_float_bytes::Int32; device_float_bytes::Int32
    kernel::Int32; viscosity::Int32; density_diffusion::Int32; mdbc::Int32; device::Int32
    shifting::Int32; kernel_output::Int32
    n_particles::Int64; max_cells::Int64
    rho0::Float64; dx::Float64; m0::Float64; alpha::Float64; g::Float64; c0::Float64; gamma::Float64
    delta_phi::Float64; CFL::Float64; Cb::Float64; nu0::Float64
    k::Float64; h::Float64; h_inv::Float64; H::Float64; H_inv::Float64; H2::Float64; alphaD::Float64; eta2::Float64
    blin_constant::Float64; smagorinsky_constant::Float64; cubic_eps::Float64
end

mutable struct SphmiProgress
    iteration::Int64; steps_done::Int64; n_rebuilds::Int64; index_counter::Int64
    total_time::Float64; last_dt::Float64; delta_x::Float64
    SphmiProgress() = new(0, 0, 0, 0, 0.0, 0.0, 0.0)
end

# model tags the engine implements; anything else falls back to the stock CPU path
visc_tag(::ZeroViscosity) = Int32(0)
visc_tag(::ArtificialViscosity) = Int32(1)
visc_tag(::Laminar) = Int32(2)
visc_tag(::LaminarSPS) = Int32(3)
visc_tag(::SPHViscosity) = nothing                      # user-defined compute_viscosity methods
ddt_tag(::ZeroGravityLinearDensityDiffusion) = Int32(1)
ddt_tag(::LinearDensityDiffusion) = Int32(2)
ddt_tag(::ComplexDensityDiffusion) = Int32(3)
ddt_tag(::SPHDensityDiffusion) = nothing               # ZeroDensityDiffusion cannot run in the reference either

function check(h::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:sphmi_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
    error("libsphmi status $rc: $msg")
end

function RunSimulationMI355X(; SimGeometry, SimMetaData::SimulationMetaData{D,T,S,K,B,L}, SimConstants, SimKernel,
                             SimLogger, SimParticles::StructArray, SimViscosity, SimDensityDiffusion,
                             ParticleNormalsPath = nothing, DeviceFloatBytes::Int = 4, Device::Int = 0) where {D,T,S,K,B,L}
    vt, dt_ = visc_tag(SimViscosity), ddt_tag(SimDensityDiffusion)
    kt = SimKernel.kernel isa WendlandC2 ? Int32(0) : (SimKernel.kernel isa CubicSpline ? Int32(1) : nothing)
    if vt === nothing || dt_ === nothing || kt === nothing
        @warn "model combination not implemented by libsphmi — running the reference CPU path"
        return RunSimulation(; SimGeometry, SimMetaData, SimConstants, SimKernel, SimLogger, SimParticles,
                             SimViscosity, SimDensityDiffusion, ParticleNormalsPath)
    end
    # the part of RunSimulation before the loop that touches particle data (src/SPHCellList.jl:827)
    SPHExample.SPHCellList.LoadMDBCNormals!(SimMetaData, SimParticles, ParticleNormalsPath)

    N = length(SimParticles)
    cfg = SphmiConfig(sizeof(SphmiConfig), 2, D, sizeof(T), DeviceFloatBytes, kt, vt, dt_, B <: SimpleMDBC ? 1 : 0,
                      Device, S <: PlanarShifting ? 1 : 0, K <: StoreKernelOutput ? 1 : 0, N, 0,
                      SimConstants.ρ₀, SimConstants.dx, SimConstants.m₀, SimConstants.α, SimConstants.g, SimConstants.c₀,
                      SimConstants.γ, SimConstants.δᵩ, SimConstants.CFL, SimConstants.Cb, SimConstants.ν₀,
                      SimKernel.k, SimKernel.h, SimKernel.h⁻¹, SimKernel.H, SimKernel.H⁻¹, SimKernel.H², SimKernel.αD, SimKernel.η²,
                      SimConstants.BlinConstant, SimConstants.SmagorinskyConstant,
                      SimKernel.kernel isa CubicSpline ? Float64(SimKernel.kernel.eps) : 0.0)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:sphmi_create, LIB), Cint, (Ref{SphmiConfig}, Ref{Ptr{Cvoid}}), cfg, href)
    rc == 0 || error("sphmi_create: " * unsafe_string(ccall((:sphmi_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    h = href[]
    try
        P = SimParticles
        typ = Vector{UInt8}(UInt8.(P.Type))                 # @enum ParticleType::UInt8
        GC.@preserve P typ begin
            check(h, ccall((:sphmi_upload, LIB), Cint,
                           (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int64}, Ptr{UInt64}, Ptr{Cvoid}),
                           h, pointer(P.Position), pointer(P.Velocity), pointer(P.Acceleration), pointer(P.Density),
                           pointer(typ), pointer(P.ID), pointer(P.GroupMarker),
                           B <: SimpleMDBC ? pointer(P.GhostPoints) : C_NULL))
        end
        check(h, ccall((:sphmi_set_clock, LIB), Cint, (Ptr{Cvoid}, Int64, Float64), h, SimMetaData.Iteration, SimMetaData.TotalTime))
        for geo in SimGeometry                                                     # MotionDefinition, :846-850
            geo.Motion === nothing && continue
            dir = Float64[geo.Motion.Direction...]
            GC.@preserve dir check(h, ccall((:sphmi_set_motion, LIB), Cint, (Ptr{Cvoid}, UInt64, Float64, Float64, Float64, Ptr{Float64}),
                                            h, UInt64(geo.GroupMarker), Float64(geo.Motion.Velocity), Float64(geo.Motion.StartTime),
                                            Float64(geo.Motion.Duration), pointer(dir)))
        end

        output = SetupVTKOutput(SimMetaData, SimParticles, SimKernel, D)          # :846
        SimMetaData.OutputIterationCounter = 1                                     # :849
        output.save_particles(SimMetaData.OutputIterationCounter)
        # (a writer that appends to the VTKHDF datasets itself can ask for the padded point layout instead of running
        #  to_3d! on the host: ccall((:sphmi_set_output_components, LIB), Cint, (Ptr{Cvoid}, Cint), h, 3) → n×3 vectors)
        # the StructArray's columns receive every output: page-lock them once (released by sphmi_destroy)
        for col in (P.Position, P.Velocity, P.Acceleration, P.Density, P.Pressure, P.ID, P.GroupMarker)
            ccall((:sphmi_host_register, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), h, pointer(col), sizeof(col))
        end
        prog = SphmiProgress()
        cells = Vector{Int64}(undef, N * D)
        while true                                                                 # :881
            t_next = SPHExample.SPHCellList.next_output_time(SimMetaData)           # :687-698
            check(h, ccall((:sphmi_advance, LIB), Cint, (Ptr{Cvoid}, Float64, Int64, Ref{SphmiProgress}), h, t_next, -1, prog))   # ≙ :883
            SimMetaData.Iteration       = prog.iteration
            SimMetaData.CurrentTimeStep = prog.last_dt
            SimMetaData.TotalTime       = prog.total_time
            SimMetaData.IndexCounter    = prog.index_counter
            SimMetaData.OutputIterationCounter += 1                                # :888
            GC.@preserve P typ cells begin                                          # state back for the VTKHDF writer
                check(h, ccall((:sphmi_download, LIB), Cint,
                               (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}, Ptr{UInt64}, Ptr{Cvoid}, Ptr{Int64}),
                               h, pointer(P.Position), pointer(P.Velocity), pointer(P.Acceleration), pointer(P.Density),
                               pointer(P.Pressure), pointer(P.ID), pointer(typ), pointer(P.GroupMarker),
                               B <: SimpleMDBC ? pointer(P.GhostPoints) : C_NULL, pointer(cells)))
                if K <: StoreKernelOutput
                    check(h, ccall((:sphmi_download_kernel_output, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                                   h, pointer(P.Kernel), pointer(P.KernelGradient)))
                end
            end
            @inbounds for i in 1:N
                P.Type[i] = ParticleType(typ[i])
                P.Cells[i] = CartesianIndex(ntuple(d -> Int(cells[(i - 1) * D + d]), D))
                # the derived per-particle flags travel with the (re-sorted) type — src/PreProcess.jl:78-100
                P.GravityFactor[i] = typ[i] == 1 ? -one(T) : (typ[i] == 3 ? one(T) : zero(T))
                P.MotionLimiter[i] = typ[i] == 1 ? one(T) : zero(T)
                P.BoundaryBool[i]  = typ[i] == 1 ? 0x00 : 0x01
            end
            output.save_particles(SimMetaData.OutputIterationCounter)              # :892
            if SimMetaData.TotalTime > SimMetaData.SimulationTime                   # :909
                output.close_files()
                break
            end
        end
    finally
        ccall((:sphmi_destroy, LIB), Cint, (Ptr{Cvoid},), h)
    end
    return nothing
end

end # module
