# main_parking.jl -- the flow of AutonomousParking/main.jl (scenario tables :43-213, warm start :216-252, ParkingDist :258, ParkingSignedDist :269,
# summary :281-286) on top of the HIP drop-in (julia/OBCAHip.jl).  Julia >= 1.6; no plots.  NOT executed in this repository's build environment
# (no Julia there); the Python twin examples/main_parking.py runs the same sequence through the same C entry points.
#
#   julia julia/main_parking.jl [backwards|parallel]
#
# Differences from main.jl, all upstream of the hot path: the Hybrid A* search is libobca_plan.so's (include/obca_plan.h) instead of
# hybrid_a_star.jl (which is Julia-0.6 code), and the path is resampled uniformly in arc length to N+1 stages instead of down-sampled by sampleN.
include(joinpath(@__DIR__, "OBCAHip.jl"))
using .OBCAHip
using Printf

const PLAN = get(ENV, "OBCA_PLAN_LIBRARY", joinpath(@__DIR__, "..", "obca_amd", "csrc", "libobca_plan.so"))

"obstHrep.jl:31-102 restated: clock-wise vertices -> stacked half-space rows A p <= b (one row per edge)"
function obstHrep(nOb, vOb, lOb)
    A = zeros(0, 2); b = zeros(0)
    for i in 1:nOb, j in 1:(vOb[i] - 1)
        v1 = lOb[i][j]; v2 = lOb[i][j + 1]
        if v1[1] == v2[1]
            a, bb = v2[2] < v1[2] ? ([1.0 0.0], v1[1]) : ([-1.0 0.0], -v1[1])
        elseif v1[2] == v2[2]
            a, bb = v1[1] < v2[1] ? ([0.0 1.0], v1[2]) : ([0.0 -1.0], -v1[2])
        else
            s, c = [v1[1] 1.0; v2[1] 1.0] \ [v1[2], v2[2]]
            a, bb = v1[1] < v2[1] ? ([-s 1.0], c) : ([s -1.0], -c)
        end
        A = vcat(A, a); push!(b, bb)
    end
    return A, b
end

"Hybrid A* through the C ABI of libobca_plan.so: returns path (K x 3), dir (K)"
function hybrid_astar(x0, xF, vObMPC, A, b, ego, L, XYbounds; opts=Float64[])      # opts: the first length(opts) options of include/obca_plan.h (empty = defaults)
    cap = 20000; path = zeros(3, cap); dir = zeros(Cint, cap); o = Float64.(vec(opts))
    n = ccall((:obca_plan_hybrid_astar2, PLAN), Cint,
              (Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cint}, Cint, Ptr{Cint}),
              Float64.(x0[1:3]), Float64.(xF[1:3]), length(vObMPC), Cint.(vec(vObMPC)), vec(permutedims(Float64.(A))), Float64.(b), Float64.(ego), L,
              Float64.([XYbounds[1], XYbounds[2], XYbounds[3], XYbounds[4]]), o, length(o), path, dir, cap, C_NULL)
    n > 0 || error("Hybrid A*: no path ($n)")
    return permutedims(path[:, 1:n]), dir[1:n]
end

"path -> (Ts, xWS (N+1)x4, uWS Nx2): uniform resampling in arc length, speed +-v_nom (0 at the ends and at direction switches), steering from the curvature"
function warm_start(path, dir, N, xF, L; v_nom=0.5)
    P = copy(path)
    for k in 2:size(P, 1)                                     # unwrap the heading (SURVEY Q16)
        d = P[k, 3] - P[k - 1, 3]; P[k, 3] = P[k - 1, 3] + mod(d + pi, 2pi) - pi
    end
    P[end, 1:2] = xF[1:2]; P[end, 3] = P[end - 1, 3] + mod(xF[3] - P[end - 1, 3] + pi, 2pi) - pi
    seg = hypot.(diff(P[:, 1]), diff(P[:, 2])); cum = vcat(0.0, cumsum(seg)); tot = cum[end]
    ss = range(0, tot, length=N + 1)
    interp(col) = [begin i = clamp(searchsortedlast(cum, s), 1, length(cum) - 1); a = seg[i] > 0 ? (s - cum[i]) / seg[i] : 0.0; P[i, col] + a * (P[i + 1, col] - P[i, col]) end for s in ss]
    X, Y, yaw = interp(1), interp(2), interp(3)
    d = [Float64(dir[clamp(searchsortedfirst(cum, s), 2, length(cum))]) for s in ss]; d[1] = dir[min(2, end)]
    Ts = tot / (N * v_nom)
    v = d .* v_nom; v[1] = 0; v[end] = 0
    for i in 2:N; d[i] != d[i + 1] && (v[i] = 0); end
    a = clamp.(diff(v) ./ Ts, -0.4, 0.4)
    ds = max.(diff(collect(ss)), 1e-9) .* [x == 0 ? 1.0 : x for x in d[2:end]]
    delta = clamp.(atan.(L .* diff(yaw) ./ ds), -0.6, 0.6)
    return Ts, hcat(X, Y, yaw, v), hcat(delta, a)
end

function main(scenario="backwards"; N=80)
    fixTime = 0; L = 2.7; ego = [3.7, 1, 1, 1]                                      # main.jl:40,63,73
    if scenario == "backwards"                                                      # main.jl:99-108
        nOb = 3; vOb = [3 3 2]
        lOb = [[[-20, 5], [-1.3, 5], [-1.3, -5]], [[1.3, -5], [1.3, 5], [20, 5]], [[20, 11], [-20, 11]]]
        xF = [0 1.3 pi/2 0]; v_nom = 0.5
    else                                                                            # main.jl:151-162
        nOb = 4; vOb = [3 3 2 2]
        lOb = [[[-20, 5], [-3, 5], [-3, 0]], [[3, 0], [3, 5], [20, 5]], [[-3, 2.5], [3, 2.5]], [[20, 11], [-20, 11]]]
        xF = [-1.35 4 0 0]; v_nom = 0.25
    end
    vObMPC = vOb .- 1                                                               # main.jl:101
    XYbounds = [-15, 15, 1, 10]; x0 = [-6 9.5 0.0 0.0]                             # main.jl:210,213
    AOb, bOb = obstHrep(nOb, vOb, lOb)                                              # main.jl:252
    t0 = time(); path, dir = hybrid_astar(x0, xF, vObMPC, AOb, bOb, ego, L, XYbounds); timeHybAstar = time() - t0
    Ts, xWS, uWS = warm_start(path, dir, N, xF, L; v_nom=v_nom); xWS[1, :] = x0
    rx, ry, ryaw = xWS[:, 1], xWS[:, 2], xWS[:, 3]
    global ego_global = ego
    println("Parking using Distance Approach (A* warm start)")                     # main.jl:256-264
    xp20, up20, scaleTime20, exitflag20, time20, lp20, np20 = OBCAHip.ParkingDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vObMPC, AOb, bOb, rx, ry, ryaw, fixTime, xWS, uWS)
    println(exitflag20 == 1 ? "  --> Distance: SUCCESSFUL." : "  --> WARNING: Problem could not be solved.")
    println("Parking using Signed Distance Approach (A* warm start)")              # main.jl:267-277
    xp10, up10, scaleTime10, exitflag10, time10, lp10, np10 = OBCAHip.ParkingSignedDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vObMPC, AOb, bOb, rx, ry, ryaw, fixTime, xWS, uWS)
    println(exitflag10 == 1 ? "  --> Signed Distance: SUCCESSFUL." : "  --> WARNING: Problem could not be solved.")
    println("********************* summary *********************")                 # main.jl:281-286
    @printf("  Time Hybrid A*: %.4f s\n  Time Distance approach: %.4f s\n  Time Signed Distance approach: %.4f s\n", timeHybAstar, time20, time10)
    @printf("  final pose error: %.2e   timeScale: %.4f\n", maximum(abs.(xp10[:, end] .- vec(xF))), scaleTime10[1])
    return xp10, up10, scaleTime10, exitflag10
end

if abspath(PROGRAM_FILE) == @__FILE__
    main(length(ARGS) >= 1 ? ARGS[1] : "backwards")
end
