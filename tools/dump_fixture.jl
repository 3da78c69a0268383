# tools/dump_fixture.jl — pins the CPU oracle (oracle/sph_oracle.c) and the engine to the REAL reference.
#
# Runs the UNMODIFIED SPHExample (RunSimulation, src/SPHCellList.jl:808-930) on the shipped layouts for one short output
# interval and writes what it leaves in SimParticles / SimMetaData — ID-sorted Position, Velocity, Density, Pressure,
# Acceleration as `<case>.csv.gz`, the loop counters as `<case>.json` — into tests/golden/reference/.
# tests/test_reference_fixtures.py checks the oracle (always) and the HIP engine (-m gpu) against these files and skips
# while they are absent: the build image has no Julia, so this script has never been run there.
#
#   cd <SPHExample checkout>            # the input/ paths below are relative to it, as in example/*.jl
#   julia -t 1 --project=. <repo>/tools/dump_fixture.jl <repo>/tests/golden/reference
#
# One thread: the reference's summation order depends on the thread count (SURVEY.md §8a Q8); the tolerance of the
# consuming test (1e-9 relative) does not need it, bit-for-bit comparisons would.
# Harness shape: test/runtests.jl:18-75 (build the structures by hand, run, inspect SimParticles).
using SPHExample, StaticArrays, CSV, Printf
import CodecZlib                     # ] add CodecZlib in the environment that runs this script (gzip for the tables)

const OUT = length(ARGS) >= 1 ? ARGS[1] : "reference_fixtures"
mkpath(OUT)

function dump(case, P, M, D)
    order = sortperm(P.ID)
    open(CodecZlib.GzipCompressorStream, joinpath(OUT, case * ".csv.gz"), "w") do io
        cols = ["ID"; ["x$d" for d in 1:D]; ["v$d" for d in 1:D]; ["a$d" for d in 1:D]; "rho"; "p"; "type"]
        println(io, join(cols, ","))
        for i in order
            vals = Any[P.ID[i]]
            append!(vals, P.Position[i]); append!(vals, P.Velocity[i]); append!(vals, P.Acceleration[i])
            push!(vals, P.Density[i]); push!(vals, P.Pressure[i]); push!(vals, Int(P.Type[i]))
            println(io, join((v isa AbstractFloat ? @sprintf("%.17g", v) : string(v) for v in vals), ","))
        end
    end
    open(joinpath(OUT, case * ".json"), "w") do io
        @printf(io, "{\"case\": \"%s\", \"n\": %d, \"dims\": %d, \"iteration\": %d, \"total_time\": %.17g, \"last_dt\": %.17g, \"index_counter\": %d, \"t_target\": %.17g, \"threads\": %d, \"version\": \"%s\"}\n",
                case, length(P), D, M.Iteration, M.TotalTime, M.CurrentTimeStep, M.IndexCounter, M.SimulationTime, Threads.nthreads(), string(pkgversion(SPHExample)))
    end
    println("wrote $case: N = $(length(P)), iterations = $(M.Iteration), t = $(M.TotalTime)")
end

# one output interval of `t_end` seconds: SimulationTime = OutputTimes = t_end ⇒ exactly one SimulationLoop call (:883)
function run_case(case; D, consts, kernel, geometry, SMode = NoShifting, BMode = NoMDBC, visc = ArtificialViscosity(),
                  ddt = LinearDensityDiffusion(), normals = nothing, t_end)
    T = Float64
    P = AllocateDataStructures(geometry)
    dir = mktempdir()
    M = SimulationMetaData{D,T,SMode,NoKernelOutput,BMode,NoLog}(SimulationName = case, SaveLocation = dir,
        SimulationTime = t_end, OutputTimes = t_end, VisualizeInParaview = false, ExportSingleVTKHDF = true,
        ExportGridCells = false, OpenLogFile = false)
    logger = SimulationLogger(dir; to_console = false)
    RunSimulation(SimGeometry = geometry, SimMetaData = M, SimConstants = consts, SimKernel = kernel, SimLogger = logger,
                  SimParticles = P, SimViscosity = visc, SimDensityDiffusion = ddt, ParticleNormalsPath = normals)
    dump(case, P, M, D)
end

geo(D, file, marker, type; motion = nothing) = Geometry{D,Float64}(CSVFile = file, GroupMarker = marker, Type = type, Motion = motion)

# StillWedgeMDBC — example/StillWedgeMDBC.jl:7-72 (BASELINE config 5); ≈25 steps
let c = SimulationConstants{Float64}(dx = 0.02, c₀ = 42.48576250492629, δᵩ = 0.1, CFL = 0.5)
    run_case("still_wedge_mdbc"; D = 2, consts = c, kernel = SPHKernelInstance{2,Float64}(WendlandC2(); dx = c.dx), BMode = SimpleMDBC,
             geometry = [geo(2, "./input/still_wedge/StillWedge_Dp0.02_Bound.csv", 1, Fixed), geo(2, "./input/still_wedge/StillWedge_Dp0.02_Fluid.csv", 2, Fluid)],
             normals = "./input/still_wedge_mdbc/StillWedge_Dp0.02_GhostNodes_Correct.csv", t_end = 0.005)
end
# Dambreak2dMDBC — example/Dambreak2dMDBC.jl:7,30-36,74-81 (dx = 0.01 with the Dp0.02 layouts, as the script has it)
let c = SimulationConstants{Float64}(dx = 0.01, c₀ = 88.14487860902641, δᵩ = 0.1, CFL = 0.5, α = 0.01)
    run_case("dam_break_2d_mdbc"; D = 2, consts = c, kernel = SPHKernelInstance{2,Float64}(WendlandC2(); dx = c.dx), BMode = SimpleMDBC,
             geometry = [geo(2, "./input/dam_break_2d/DamBreak2d_Dp0.02_MDBC_Bound_ThreeLayers.csv", 1, Fixed), geo(2, "./input/dam_break_2d/DamBreak2d_Dp0.02_MDBC_Fluid_ThreeLayers.csv", 2, Fluid)],
             normals = "./input/dam_break_2d/DamBreak2d_Dp0.02_MDBC_GhostNodes_ThreeLayers.csv", t_end = 0.002)
end
# 2-D dam break without mDBC (BASELINE configs 1/2; parameters of SURVEY.md §8d C1)
let c = SimulationConstants{Float64}(dx = 0.02, c₀ = 88.14487860902641, δᵩ = 0.1, CFL = 0.2, α = 0.01)
    run_case("dam_break_2d"; D = 2, consts = c, kernel = SPHKernelInstance{2,Float64}(WendlandC2(); dx = c.dx),
             geometry = [geo(2, "./input/dam_break_2d/DamBreak2d_Dp0.02_Bound.csv", 1, Fixed), geo(2, "./input/dam_break_2d/DamBreak2d_Dp0.02_Fluid.csv", 2, Fluid)],
             t_end = 0.003)
end
# Dambreak3d at the shipped Dp0.02 — example/Dambreak3d.jl:8-59 with dx = 0.02
let dx = 0.02, c = SimulationConstants{Float64}(dx = dx, c₀ = 33.14, α = 0.1, m₀ = 1000 * dx^3, CFL = 0.2)
    run_case("dam_break_3d_dp0.02"; D = 3, consts = c, kernel = SPHKernelInstance{3,Float64}(WendlandC2(); h = 1 * sqrt(3 * dx^2)),
             geometry = [geo(3, "./input/dam_break_3d/DamBreak3d_Dp0.02_Bound.csv", 1, Fixed), geo(3, "./input/dam_break_3d/DamBreak3d_Dp0.02_Fluid.csv", 2, Fluid)],
             t_end = 0.005)
end
# MovingSquare2d at the shipped Dp0.04 — example/MovingSquare2d.jl:9-79 (moving body + LaminarSPS + PlanarShifting, k = √2)
let c = SimulationConstants{Float64}(dx = 0.04, c₀ = 28, δᵩ = 0.1, g = 0, Cb = 112000, α = 1e-6, CFL = 0.2)
    m = MotionDetails{2,Float64}(Velocity = 2.8, StartTime = 0.0, Duration = 3.0, Direction = SVector{2,Float64}(1.0, 0.0))
    run_case("moving_square_2d_dp0.04"; D = 2, consts = c, kernel = SPHKernelInstance{2,Float64}(WendlandC2(); dx = c.dx, k = sqrt(2)),
             SMode = PlanarShifting, visc = LaminarSPS(),
             geometry = [geo(2, "./input/moving_square_2d/MovingSquare_Dp0.04_Fixed.csv", 1, Fixed), geo(2, "./input/moving_square_2d/MovingSquare_Dp0.04_Fluid.csv", 2, Fluid),
                         geo(2, "./input/moving_square_2d/MovingSquare_Dp0.04_Square.csv", 3, Moving; motion = m)],
             t_end = 0.01)
end
