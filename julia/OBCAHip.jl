# OBCAHip.jl -- thin Julia shim (Julia >= 1.6) over libobca_hip.so.
#
# Drop-in replacements, with the reference's positional signatures and return tuples, for
#   ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)   (AutonomousParking/ParkingSignedDist.jl:29)
#   DualMultWS(N,nOb,vOb,A,b,rx,ry,ryaw)                                                   (AutonomousParking/DualMultWS.jl:29)
#   ParkingDist(...)                                                                       (AutonomousParking/ParkingDist.jl:29)
#   QuadcopterSignedDist(x0,xF,N,Ts,R,ob1,ob2,ob3,ob4,ob5,xWS,uWS,timeWS)                  (QuadcopterNavigation/QuadcopterSignedDist.jl:25)
#   QuadcopterDist(...)                                                                    (QuadcopterNavigation/QuadcopterDist.jl:25)
# plus batched variants and multi-GPU contexts.  Julia arrays are column-major, which is exactly the "stage-contiguous" layout of the C ABI
# (include/obca_hip.h), so every array is passed with zero copies.
#
# NOTE: Julia is not installed in the build environment of this repository, so this file has not been executed there; the
# same C entry points are exercised through ctypes by tests/test_gpu_parity.py.
module OBCAHip

const LIB = get(ENV, "OBCA_HIP_LIBRARY", joinpath(@__DIR__, "..", "obca_amd", "csrc", "libobca_hip.so"))
# One hardware queue per stream: the HIP runtime's default of four serialises streams that share one (the worker lanes of the host-pointer entry points, several contexts
# in flight).  Read by the runtime at its first call in the process; a value the caller has set stays.  (INTEGRATION.md, profiles/r06_hw_queues.txt)
function __init__()
    haskey(ENV, "GPU_MAX_HW_QUEUES") || (ENV["GPU_MAX_HW_QUEUES"] = "16")
end

mutable struct Context
    h::Ptr{Cvoid}
end

function Context(device::Integer=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:obca_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint), r, device)
    rc == 0 || error("obca_create failed: " * unsafe_string(ccall((:obca_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    c = Context(r[])
    finalizer(x -> ccall((:obca_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), c)
    return c
end

"Context over several GPUs of the node (all visible ones by default): the batched calls shard their batch over them (obca_create_multi)."
function MultiContext(devices::Vector{<:Integer}=Int[])
    r = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:obca_create_multi, LIB), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cint}, Cint), r, isempty(devices) ? C_NULL : Cint.(devices), length(devices))
    rc == 0 || error("obca_create_multi failed: " * unsafe_string(ccall((:obca_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    c = Context(r[])
    finalizer(x -> ccall((:obca_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), c)
    return c
end
device_count(c::Context) = Int(ccall((:obca_device_count, LIB), Cint, (Ptr{Cvoid},), c.h))
"make every later call of this module use context `c` (e.g. `use!(MultiContext())` for all GPUs of the node)"
use!(c::Context) = (_ctx[] = c)

const _ctx = Ref{Union{Nothing,Context}}(nothing)
ctx() = (_ctx[] === nothing && (_ctx[] = Context(0)); _ctx[])
lasterr(c) = unsafe_string(ccall((:obca_last_error, LIB), Cstring, (Ptr{Cvoid},), c.h))

f64(a) = convert(Array{Float64}, a)

"""
Interior-point options: the record `obca_opts` of include/obca_hip.h, field for field.  `ipopt_opts()` = the reference's IPOPT configuration as far as the kernels carry it
(ParkingSignedDist.jl:41-43 incl. recalc_y = "yes", IPOPT's default second-order correction max_soc = 4 and least-squares initial multipliers): the default of the drop-ins
`ParkingSignedDist` / `ParkingDist`.  `default_opts()` = the library's throughput defaults (the three switches off: the same solved set, a quarter fewer GPU seconds, 1-18 % of the
instances of a batch end in another local solution -- include/obca_hip.h has the numbers): the default of the batched calls (`opts=nothing`).
"""
mutable struct Opts
    tol::Cdouble; max_iter::Cint
    mu_init::Cdouble; kappa_eps::Cdouble; kappa_mu::Cdouble; theta_mu::Cdouble; tau_min::Cdouble; bound_push::Cdouble; bound_frac::Cdouble
    dw_min::Cdouble; dw0::Cdouble; dw_max::Cdouble; kw_inc0::Cdouble; kw_inc::Cdouble; kw_dec::Cdouble; dc_bar::Cdouble; kappa_c::Cdouble
    gamma_theta::Cdouble; gamma_phi::Cdouble; delta::Cdouble; s_theta::Cdouble; s_phi::Cdouble; eta_phi::Cdouble; gamma_alpha::Cdouble; s_max::Cdouble; kappa_sigma::Cdouble
    constr_viol_tol::Cdouble; dual_inf_tol::Cdouble; compl_inf_tol::Cdouble; rho_term::Cdouble
    max_soc::Cint; recalc_y::Cint; lsq_init::Cint; obj_scaling::Cint; restoration::Cint
    Opts() = new()
end
function default_opts()
    o = Opts()
    ccall((:obca_default_opts, LIB), Cint, (Ref{Opts},), o) == 0 || error("obca_default_opts failed")
    return o
end
function ipopt_opts()
    o = Opts()
    ccall((:obca_reference_opts, LIB), Cint, (Ref{Opts},), o) == 0 || error("obca_reference_opts failed")
    return o
end
optsptr(o) = o === nothing ? C_NULL : pointer_from_objref(o)
"the reference's IPOPT configuration of the quadcopter call as far as the kernel carries it (max_soc = 4, least-squares initial multipliers, gradient-based objective scaling; recalc_y = \"no\" as QuadcopterSignedDist.jl:29 sets it): default of the quadcopter drop-ins"
function quadcopter_ipopt_opts()
    o = Opts()
    ccall((:obca_quadcopter_reference_opts, LIB), Cint, (Ref{Opts},), o) == 0 || error("obca_quadcopter_reference_opts failed")
    return o
end

"""
    ParkingSignedDist_batch(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS)

Batched form: x0, xF are 4xB; rx, ry, ryaw (N+1)xB; xWS 4x(N+1)xB (already transposed to the x layout); uWS 2xNxB; Ts a vector of
length B; the obstacle set (nOb, vOb, A (Mx2), b) is shared by the batch.  Returns (xp 4x(N+1)xB, up 2xNxB, timeScale (N+1)xB,
exitflag B, time, lp Mx(N+1)xB, np 4nObx(N+1)xB).
"""
function ParkingSignedDist_batch(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS; opts=nothing)
    B = size(x0, 2); M = sum(vOb)
    nObs = fill(Cint(nOb), B); vflat = repeat(Cint.(vec(vOb)), B)
    At = repeat(vec(permutedims(f64(A))), B)         # row k of A as (A[k,1], A[k,2]), per instance
    bt = repeat(vec(f64(b)), B)
    xp = zeros(4, N + 1, B); up = zeros(2, N, B); ts = zeros(N + 1, B); ef = zeros(Cint, B)
    lp = zeros(M, N + 1, B); np = zeros(4nOb, N + 1, B); info = zeros(8, B)
    t0 = time()
    rc = GC.@preserve opts ccall((:obca_parking_signed_dist_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, B, N, f64(vec(Ts)), L, f64(vec(ego)), f64(vec(XYbounds)), fixTime, f64(x0), f64(xF), nObs, vflat, At, bt,
               f64(rx), f64(ry), f64(ryaw), f64(xWS), f64(uWS), C_NULL, C_NULL,   # lWS = nWS = NULL: DualMultWS runs on the GPU
               optsptr(opts), xp, up, ts, ef, lp, np, C_NULL, info)
    rc == 0 || error("obca_parking_signed_dist_batch failed: " * lasterr(ctx()))
    return xp, up, ts, ef, time() - t0, lp, np
end

"Drop-in for ParkingSignedDist.jl:29 (one instance): same arguments, same 7-tuple."
function ParkingSignedDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS; opts=ipopt_opts())      # default: the reference's IPOPT configuration; `default_opts()` = the library's throughput defaults
    xp, up, ts, ef, t, lp, np = ParkingSignedDist_batch(reshape(f64(vec(x0)), 4, 1), reshape(f64(vec(xF)), 4, 1), N, [Float64(Ts)], L, ego,
        XYbounds, nOb, vOb, A, b, reshape(f64(rx)[1:N+1], N + 1, 1), reshape(f64(ry)[1:N+1], N + 1, 1), reshape(f64(ryaw)[1:N+1], N + 1, 1),
        fixTime, reshape(permutedims(f64(xWS)[1:N+1, :]), 4, N + 1, 1), reshape(permutedims(f64(uWS)[1:N, :]), 2, N, 1); opts=opts)
    timeScalep = fixTime == 1 ? ones(1, N + 1) : ts[:, 1]          # ParkingSignedDist.jl:304-308
    return xp[:, :, 1], up[:, :, 1], timeScalep, Int(ef[1]), t, lp[:, :, 1], np[:, :, 1]
end

"Drop-in for ParkingDist.jl:29 (collision-free sibling; entry point obca_parking_dist_batch has the same arguments minus the slack output)."
function ParkingDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS; opts=ipopt_opts())      # (ParkingDist.jl:41 sets recalc_y = "yes" too)
    M = sum(vOb)
    xp = zeros(4, N + 1); up = zeros(2, N); ts = zeros(N + 1); ef = zeros(Cint, 1); lp = zeros(M, N + 1); np = zeros(4nOb, N + 1)
    t0 = time()
    rc = GC.@preserve opts ccall((:obca_parking_dist_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, 1, N, [Float64(Ts)], L, f64(vec(ego)), f64(vec(XYbounds)), fixTime, f64(vec(x0)), f64(vec(xF)), Cint[nOb], Cint.(vec(vOb)),
               vec(permutedims(f64(A))), f64(vec(b)), f64(rx)[1:N+1], f64(ry)[1:N+1], f64(ryaw)[1:N+1], vec(permutedims(f64(xWS)[1:N+1, :])),
               vec(permutedims(f64(uWS)[1:N, :])), C_NULL, C_NULL, optsptr(opts), xp, up, ts, ef, lp, np, C_NULL)
    rc == 0 || error("obca_parking_dist_batch failed: " * lasterr(ctx()))
    timeScalep = fixTime == 1 ? ones(1, N + 1) : ts
    return xp, up, timeScalep, Int(ef[1]), time() - t0, lp, np          # ParkingDist.jl:313
end

"Drop-in for DualMultWS.jl:29; `ego` defaults to the global the reference reads (DualMultWS.jl:39-45). Returns (lp (N+1)xM, np (N+1)x4nOb)."
function DualMultWS(N, nOb, vOb, A, b, rx, ry, ryaw; ego=Main.ego)
    M = sum(vOb)
    lw = zeros(M, N + 1); nw = zeros(4nOb, N + 1)
    rc = ccall((:obca_dualmult_ws_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, 1, N, f64(vec(ego)), Cint[nOb], Cint.(vec(vOb)), vec(permutedims(f64(A))), f64(vec(b)), f64(rx)[1:N+1], f64(ry)[1:N+1],
               f64(ryaw)[1:N+1], lw, nw, C_NULL)
    rc == 0 || error("obca_dualmult_ws_batch failed: " * lasterr(ctx()))
    return permutedims(lw), permutedims(nw)                        # DualMultWS.jl:81-84 returns the transposes
end


"""
    QuadcopterSignedDist_batch(x0, xF, N, Ts, R, ob, xWS, timeWS; dual_ws=true, dist=false)

Batched form: x0, xF 12xB; Ts, timeWS vectors of length B; ob 6x5xB (ob1..ob5 of every instance back to back, each
[xmax,ymax,zmax,-xmin,-ymin,-zmin]); xWS 12x(N+1)xB.  Returns (xp 12x(N+1)xB, up 4xNxB, timeScale (N+1)xB, exitflag B, time, lp 30x(N+1)xB,
status codes B).  dist=true solves the QuadcopterDist formulation (obca_quadcopter_dist_batch).  opts=nothing: the library's throughput defaults
(obca_quadcopter_default_opts); `quadcopter_ipopt_opts()`: with IPOPT's second-order correction and least-squares initial multipliers.
"""
function QuadcopterSignedDist_batch(x0, xF, N, Ts, R, ob, xWS, timeWS; dual_ws::Bool=true, dist::Bool=false, opts=nothing)
    B = size(x0, 2)
    xp = zeros(12, N + 1, B); up = zeros(4, N, B); ts = zeros(N + 1, B); ef = zeros(Cint, B); lp = zeros(30, N + 1, B); info = zeros(8, B)
    t0 = time()
    if dist
        rc = GC.@preserve opts ccall((:obca_quadcopter_dist_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                    Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}),
                   ctx().h, B, N, f64(vec(Ts)), Float64(R), f64(x0), f64(xF), f64(ob), f64(xWS), C_NULL, f64(vec(timeWS)), dual_ws ? 1 : 0, optsptr(opts),
                   xp, up, ts, ef, lp, info)
    else
        rc = GC.@preserve opts ccall((:obca_quadcopter_signed_dist_batch, LIB), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                    Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                   ctx().h, B, N, f64(vec(Ts)), Float64(R), f64(x0), f64(xF), f64(ob), f64(xWS), C_NULL, f64(vec(timeWS)), dual_ws ? 1 : 0, optsptr(opts),
                   xp, up, ts, ef, lp, C_NULL, info)
    end
    rc == 0 || error("obca_quadcopter_(signed_)dist_batch failed: " * lasterr(ctx()))
    return xp, up, ts, ef, time() - t0, lp, info[1, :]
end

_quad_status(c) = c == 0 ? "Optimal" : (c == 1 ? "UserLimit" : "Error")

# xWS is 12 x (N+1) in the reference (mainQuadcopter.jl:136 builds [rx'; ry'; rz'; zeros...], QuadcopterSignedDist.jl:201 does setvalue(x, xWS)
# without a transpose): column-major, that is already the stage-contiguous layout of the C ABI.
function _quad_one(x0, xF, N, Ts, R, obs, xWS, timeWS, dual_ws, dist, opts)
    ob = reshape(f64(vcat(map(vec, obs)...)), 6, 5, 1)               # [xmax,ymax,zmax,-xmin,-ymin,-zmin] per box (:162-166)
    xp, up, ts, ef, t, lp, st = QuadcopterSignedDist_batch(reshape(f64(vec(x0)), 12, 1), reshape(f64(vec(xF)), 12, 1), N, [Float64(Ts)], R, ob,
        reshape(f64(xWS)[:, 1:N+1], 12, N + 1, 1), [Float64(timeWS)]; dual_ws=dual_ws, dist=dist, opts=opts)
    return xp[:, :, 1], up[:, :, 1], ts[:, 1], Int(ef[1]), t, lp[:, :, 1], _quad_status(st[1])
end

"Drop-in for QuadcopterSignedDist.jl:25 (one instance): same arguments, same 7-tuple (xp, up, timeScalep, exitflag, time, lp, status), :298."
QuadcopterSignedDist(x0, xF, N, Ts, R, ob1, ob2, ob3, ob4, ob5, xWS, uWS, timeWS; dual_ws::Bool=true, opts=quadcopter_ipopt_opts()) =
    _quad_one(x0, xF, N, Ts, R, (ob1, ob2, ob3, ob4, ob5), xWS, timeWS, dual_ws, false, opts)

"Drop-in for QuadcopterDist.jl:25 (call site mainQuadcopter.jl:145): the collision-free sibling, same arguments and 7-tuple (:282)."
QuadcopterDist(x0, xF, N, Ts, R, ob1, ob2, ob3, ob4, ob5, xWS, uWS, timeWS; dual_ws::Bool=true, opts=quadcopter_ipopt_opts()) =
    _quad_one(x0, xF, N, Ts, R, (ob1, ob2, ob3, ob4, ob5), xWS, timeWS, dual_ws, true, opts)

end # module
