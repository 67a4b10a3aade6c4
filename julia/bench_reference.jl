# bench_reference.jl -- times the UNTOUCHED reference (ParkingSignedDist.jl: JuMP + IPOPT) on the instances of tests/golden/reference_instances_backwards.json (or _parallel.json)
# and records what it returns, so that the first box with Julia 0.6 + JuMP 0.18 + Ipopt.jl can produce the pin this repository could not
# (SURVEY.md 8c / BASELINE.md 3.3: parity is unpinned against IPOPT).  Written for the reference's own vintage (Julia 0.5 / 0.6: `include`, `tic`/`toq`
# inside the reference, JSON.jl for I/O).  NOT executed here: neither Julia nor Ipopt exists in the build environment.
#
#   cd /path/to/OBCA/AutonomousParking && julia /path/to/repo/julia/bench_reference.jl /path/to/repo/tests/golden/reference_instances_backwards.json (or _parallel.json) out.json
#
# Input  (written by tests/golden/make_reference_instances.py): {"L":..,"ego":[..],"XYbounds":[..],"nOb":..,"vOb":[..],"A":[[..]..],"b":[..],"N":80,
#          "instances":[{"x0":[4],"xF":[4],"Ts":..,"xWS":[[4]xN+1],"uWS":[[2]xN]}, ...]}   -- the first 8 instances of BASELINE config 2 and 4 of config 3
# Output: per instance exitflag, solve time (toq() around solve(m), ParkingSignedDist.jl:239-241), xp, up, timeScale, objective recomputed from :86-92.
#         tests/test_pin_cpu.py::test_reference_output_if_present compares it with the oracle at the stated parity tolerance (objective 1e-4 relative,
#         states / inputs 1e-3, timeScale 1e-4) once the file is committed as tests/golden/reference_output_backwards.json / _parallel.json.
using JSON
include("setup.jl")                       # the reference's own setup (JuMP, Ipopt, PyPlot ...)
include("ParkingSignedDist.jl")
include("DualMultWS.jl")
include("ParkingConstraints.jl")

function objective(xp, up, ts, sl, rx, ry, ryaw, Ts, N)        # ParkingSignedDist.jl:86-92 (variable time)
    J = sum(0.01 * up[1, i]^2 + 0.1 * up[2, i]^2 for i in 1:N)
    J += sum(0.1 * ((up[1, i + 1] - up[1, i]) / (ts[i] * Ts))^2 + 0.1 * ((up[2, i + 1] - up[2, i]) / (ts[i] * Ts))^2 for i in 1:N-1)
    J += 0.1 * (up[1, 1] / (ts[1] * Ts))^2 + 0.1 * (up[2, 1] / (ts[1] * Ts))^2
    J += sum(0.5 * ts[i] + ts[i]^2 for i in 1:N+1) + sum(1e-4 * xp[4, i]^2 for i in 1:N+1)
    J += sum(1e-3 * (xp[1, i] - rx[i])^2 + 1e-3 * (xp[2, i] - ry[i])^2 + 1e-4 * (xp[3, i] - ryaw[i])^2 for i in 1:N+1)
    return J                                # the slack terms (sl is not returned by the reference, SURVEY a1) are added by the comparing test
end

inp = JSON.parsefile(ARGS[1]); out = Dict("instances" => [])
global ego = Float64.(inp["ego"])          # DualMultWS.jl:39-45 reads the global
A = hcat([Float64.(r) for r in inp["A"]]...)'; b = Float64.(inp["b"]); N = inp["N"]
for (k, q) in enumerate(inp["instances"])
    xWS = hcat([Float64.(r) for r in q["xWS"]]...)'; uWS = hcat([Float64.(r) for r in q["uWS"]]...)'
    x0 = reshape(Float64.(q["x0"]), 1, 4); xF = reshape(Float64.(q["xF"]), 1, 4)
    xp, up, ts, ef, t, lp, np = ParkingSignedDist(x0, xF, N, q["Ts"], inp["L"], ego, Float64.(inp["XYbounds"]), inp["nOb"], reshape(Int.(inp["vOb"]), 1, :), A, b,
                                                   xWS[:, 1], xWS[:, 2], xWS[:, 3], 0, xWS, uWS)
    push!(out["instances"], Dict("exitflag" => ef, "time" => t, "xp" => xp, "up" => up, "timeScale" => ts, "lp" => lp, "np" => np,
                                 "objective_without_slack" => objective(xp, up, ts, nothing, xWS[:, 1], xWS[:, 2], xWS[:, 3], q["Ts"], N)))
    println("instance $k: exitflag $ef, $(round(t, 3)) s")
end
open(ARGS[2], "w") do f; JSON.print(f, out); end
