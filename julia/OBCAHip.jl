# OBCAHip.jl -- thin Julia shim (Julia >= 1.6) over libobca_hip.so.
#
# Drop-in replacements, with the reference's positional signatures and return tuples, for
#   ParkingSignedDist(x0,xF,N,Ts,L,ego,XYbounds,nOb,vOb,A,b,rx,ry,ryaw,fixTime,xWS,uWS)   (AutonomousParking/ParkingSignedDist.jl:29)
#   DualMultWS(N,nOb,vOb,A,b,rx,ry,ryaw)                                                   (AutonomousParking/DualMultWS.jl:29)
# plus batched variants.  Julia arrays are column-major, which is exactly the "stage-contiguous" layout of the C ABI
# (include/obca_hip.h), so every array is passed with zero copies.
#
# NOTE: Julia is not installed in the build environment of this repository, so this file has not been executed there; the
# same C entry points are exercised through ctypes by tests/test_gpu_parity.py.
module OBCAHip

const LIB = get(ENV, "OBCA_HIP_LIBRARY", joinpath(@__DIR__, "..", "obca_amd", "csrc", "libobca_hip.so"))

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

const _ctx = Ref{Union{Nothing,Context}}(nothing)
ctx() = (_ctx[] === nothing && (_ctx[] = Context(0)); _ctx[])
lasterr(c) = unsafe_string(ccall((:obca_last_error, LIB), Cstring, (Ptr{Cvoid},), c.h))

f64(a) = convert(Array{Float64}, a)

"""
    ParkingSignedDist_batch(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS)

Batched form: x0, xF are 4xB; rx, ry, ryaw (N+1)xB; xWS 4x(N+1)xB (already transposed to the x layout); uWS 2xNxB; Ts a vector of
length B; the obstacle set (nOb, vOb, A (Mx2), b) is shared by the batch.  Returns (xp 4x(N+1)xB, up 2xNxB, timeScale (N+1)xB,
exitflag B, time, lp Mx(N+1)xB, np 4nObx(N+1)xB).
"""
function ParkingSignedDist_batch(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS)
    B = size(x0, 2); M = sum(vOb)
    nObs = fill(Cint(nOb), B); vflat = repeat(Cint.(vec(vOb)), B)
    At = repeat(vec(permutedims(f64(A))), B)         # row k of A as (A[k,1], A[k,2]), per instance
    bt = repeat(vec(f64(b)), B)
    xp = zeros(4, N + 1, B); up = zeros(2, N, B); ts = zeros(N + 1, B); ef = zeros(Cint, B)
    lp = zeros(M, N + 1, B); np = zeros(4nOb, N + 1, B); info = zeros(8, B)
    t0 = time()
    rc = ccall((:obca_parking_signed_dist_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, B, N, f64(vec(Ts)), L, f64(vec(ego)), f64(vec(XYbounds)), fixTime, f64(x0), f64(xF), nObs, vflat, At, bt,
               f64(rx), f64(ry), f64(ryaw), f64(xWS), f64(uWS), C_NULL, C_NULL,   # lWS = nWS = NULL: DualMultWS runs on the GPU
               C_NULL, xp, up, ts, ef, lp, np, C_NULL, info)
    rc == 0 || error("obca_parking_signed_dist_batch failed: " * lasterr(ctx()))
    return xp, up, ts, ef, time() - t0, lp, np
end

"Drop-in for ParkingSignedDist.jl:29 (one instance): same arguments, same 7-tuple."
function ParkingSignedDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS)
    xp, up, ts, ef, t, lp, np = ParkingSignedDist_batch(reshape(f64(vec(x0)), 4, 1), reshape(f64(vec(xF)), 4, 1), N, [Float64(Ts)], L, ego,
        XYbounds, nOb, vOb, A, b, reshape(f64(rx)[1:N+1], N + 1, 1), reshape(f64(ry)[1:N+1], N + 1, 1), reshape(f64(ryaw)[1:N+1], N + 1, 1),
        fixTime, reshape(permutedims(f64(xWS)[1:N+1, :]), 4, N + 1, 1), reshape(permutedims(f64(uWS)[1:N, :]), 2, N, 1))
    timeScalep = fixTime == 1 ? ones(1, N + 1) : ts[:, 1]          # ParkingSignedDist.jl:304-308
    return xp[:, :, 1], up[:, :, 1], timeScalep, Int(ef[1]), t, lp[:, :, 1], np[:, :, 1]
end

"Drop-in for ParkingDist.jl:29 (collision-free sibling; entry point obca_parking_dist_batch has the same arguments minus the slack output)."
function ParkingDist(x0, xF, N, Ts, L, ego, XYbounds, nOb, vOb, A, b, rx, ry, ryaw, fixTime, xWS, uWS)
    M = sum(vOb)
    xp = zeros(4, N + 1); up = zeros(2, N); ts = zeros(N + 1); ef = zeros(Cint, 1); lp = zeros(M, N + 1); np = zeros(4nOb, N + 1)
    t0 = time()
    rc = ccall((:obca_parking_dist_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, 1, N, [Float64(Ts)], L, f64(vec(ego)), f64(vec(XYbounds)), fixTime, f64(vec(x0)), f64(vec(xF)), Cint[nOb], Cint.(vec(vOb)),
               vec(permutedims(f64(A))), f64(vec(b)), f64(rx)[1:N+1], f64(ry)[1:N+1], f64(ryaw)[1:N+1], vec(permutedims(f64(xWS)[1:N+1, :])),
               vec(permutedims(f64(uWS)[1:N, :])), C_NULL, C_NULL, C_NULL, xp, up, ts, ef, lp, np, C_NULL)
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


"Drop-in for QuadcopterSignedDist.jl:25 (one instance): same arguments, same 7-tuple (xp, up, timeScalep, exitflag, time, lp, status)."
function QuadcopterSignedDist(x0, xF, N, Ts, R, ob1, ob2, ob3, ob4, ob5, xWS, uWS, timeWS; dual_ws::Bool=true)
    xp = zeros(12, N + 1); up = zeros(4, N); ts = zeros(N + 1); ef = zeros(Cint, 1); lp = zeros(30, N + 1); info = zeros(8)
    ob = f64(vcat(vec(ob1), vec(ob2), vec(ob3), vec(ob4), vec(ob5)))     # 6 x 5: [xmax,ymax,zmax,-xmin,-ymin,-zmin] per box (:162-166)
    t0 = time()
    rc = ccall((:obca_quadcopter_signed_dist_batch, LIB), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx().h, 1, N, [Float64(Ts)], Float64(R), f64(vec(x0)), f64(vec(xF)), ob, vec(permutedims(f64(xWS)[1:N+1, :])), C_NULL,
               [Float64(timeWS)], dual_ws ? 1 : 0, C_NULL, xp, up, ts, ef, lp, C_NULL, info)
    rc == 0 || error("obca_quadcopter_signed_dist_batch failed: " * lasterr(ctx()))
    status = info[1] == 0 ? "Optimal" : (info[1] == 1 ? "UserLimit" : "Error")
    return xp, up, ts, Int(ef[1]), time() - t0, lp, status            # QuadcopterSignedDist.jl:298
end

end # module
