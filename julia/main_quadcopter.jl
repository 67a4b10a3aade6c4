# main_quadcopter.jl -- the flow of QuadcopterNavigation/mainQuadcopter.jl (scenario :29-54, 3-D A* :116-129, Ts_as :131, warm start :134-138,
# QuadcopterDist :145, QuadcopterSignedDist :152) on top of the HIP drop-in (julia/OBCAHip.jl).  Julia >= 1.6; no plots, no constrSatisfaction.
# NOT executed in this repository's build environment (no Julia there); tests/test_gpu_quad_parity.py::test_reference_main_call_runs_as_is runs the
# same sequence through the same C entry points from Python.
#
#   julia julia/main_quadcopter.jl
#
# Differences from mainQuadcopter.jl, all upstream of the hot path: the grid search is libobca_plan.so's obca_plan_astar3d (include/obca_plan.h)
# instead of a_star_3D.jl (Julia-0.6 code; same 1.0 grid in the script's x10 units = 0.1 m), and the boxes are passed as the plot call at :146 leaves
# them (plotTrajQuadcopter.jl:27-110 clamps its obstacle arguments to the room in place, SURVEY Q3) since nothing is plotted here.
include(joinpath(@__DIR__, "OBCAHip.jl"))
using .OBCAHip
using Printf

const PLAN = get(ENV, "OBCA_PLAN_LIBRARY", joinpath(@__DIR__, "..", "obca_amd", "csrc", "libobca_plan.so"))

"3-D grid A* through the C ABI of libobca_plan.so: way-points (3 x K) from start to goal around the boxes inflated by `clear`"
function astar3d(start, goal, boxes, clear, room, res)
    cap = 4096; path = zeros(3, cap)
    n = ccall((:obca_plan_astar3d, PLAN), Cint,
              (Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cint, Ptr{Cint}),
              Float64.(start), Float64.(goal), length(boxes), vcat(map(b -> Float64.(vec(b)), boxes)...), Float64(clear), Float64.(room), Float64(res), path, cap, C_NULL)
    n > 0 || error("3-D A*: no path ($n)")
    return path[:, 1:n]
end

function main()
    egoR = 0.25                                                                     # mainQuadcopter.jl:29
    room = [10.0, 10.0, 5.0]                                                        # ob6, :44
    # :31-42, clamped to the room as plotTrajQuadcopter leaves them (Q3): [xmax, ymax, zmax, -xmin, -ymin, -zmin]
    ob12 = [2.5, 10, 5, -2, 0, -0.6]; ob22 = [7.5, 10, 5, -7, -5, 0]; ob32 = [7.5, 4, 5, -7, 0, 0]; ob42 = [7.5, 5, 2, -7, -4, 0]; ob52 = [7.5, 5, 5, -7, -4, -3]
    x0 = [1 1 3 0 0 0 0 0 0 0 0 0]; xF = [9 3 2 0 0 0 0 0 0 0 0 0]                  # :47-50
    Ts = 0.25                                                                       # :53
    t0 = time()
    r = astar3d(x0[1:3], xF[1:3], (ob12, ob22, ob32, ob42, ob52), 0.4, room, 0.1)   # :116-128: grid resolution 1.0 in the script's x10 units
    timeAstar = time() - t0
    N_as = size(r, 2) - 1                                                           # :129
    Ts_as = round((Ts * 80 / N_as) * 100) / 100                                     # :131
    xWS_as = [r; zeros(9, N_as + 1)]                                                # :134-136: 12 x (N_as + 1)
    uWS_as = 0.5 * ones(4, N_as); timeWS_as = 1                                     # :137-138
    println("Trajectory using Distance Approach (Collision Avoidance, A star)")    # :144-145
    xp1, up1, scaleTime1, exitflag1, time1, l1, status1 = OBCAHip.QuadcopterDist(x0, xF, N_as, Ts_as, egoR, ob12, ob22, ob32, ob42, ob52, xWS_as, uWS_as, timeWS_as; dual_ws=false)
    println("Trajectory using Signed Distance Approach (Minimum Penetration, A star)")   # :151-152
    xp2, up2, scaleTime2, exitflag2, time2, l2, status2 = OBCAHip.QuadcopterSignedDist(x0, xF, N_as, Ts_as, egoR, ob12, ob22, ob32, ob42, ob52, xWS_as, uWS_as, timeWS_as; dual_ws=false)
    @printf("  A*: %d way-points, %.3f s;  N = %d, Ts = %.2f\n", N_as + 1, timeAstar, N_as, Ts_as)
    @printf("  Distance:        exitflag %d (%s)  %.4f s  timeScale %.4f\n", exitflag1, status1, time1, scaleTime1[1])
    @printf("  Signed distance: exitflag %d (%s)  %.4f s  timeScale %.4f  final position error %.2e\n", exitflag2, status2, time2, scaleTime2[1], maximum(abs.(xp2[1:3, end] .- vec(xF)[1:3])))
    println("---- Done ----")
    return xp2, up2, scaleTime2, exitflag2
end

if abspath(PROGRAM_FILE) == @__FILE__
    main()
end
