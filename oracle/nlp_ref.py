"""
TEST INFRASTRUCTURE ONLY -- part of the CPU oracle, never imported by the product path.

Flat (unstructured) restatement of the reference's parking signed-distance NLP in
torch fp64, with derivatives by autograd, plus a dense textbook interior-point
solver.  It is slow (dense KKT, O(n^3)) and only meant for short horizons: it is the
independent cross-check of the structured C oracle (oracle/obca_oracle.c) and of the
HIP path -- different derivatives (autograd vs closed form), different linear algebra
(dense LDL^T vs condensed Riccati).

Follows /root/reference/AutonomousParking/ParkingSignedDist.jl:
  variables :49-59, objective :77-93, bounds :100-115, boundary :122-131,
  dynamics :139-155, steering rate :157-174, obstacle rows :182-208.
Parity status: UNPINNED (the reference ships no golden vectors and IPOPT/Julia are
not available in this environment).

Modelling notes (same feasible set / optimum as the reference):
  * timeScale[i]==timeScale[i+1] chain (:152-154) is carried as ONE scalar t; the
    N+1 copies of its bound barrier and of the cost 0.5t+t^2 are kept as a weight N+1.
  * x[:,1]==x0 is eliminated (x_0 is a constant); x[:,N+1]==xF is kept as equality.
  * two-sided steering-rate rows and the obstacle >= rows get explicit slacks.
"""
import numpy as np
import torch

torch.set_default_dtype(torch.float64)
DMIN = 0.05


class ParkingNLP:
    def __init__(self, x0, xF, N, Ts, L, ego, XYb, vOb, A, b, rx, ry, ryaw, fixTime=0, dist=0):
        self.N = N
        self.dist = int(dist)     # 1: ParkingDist.jl -- the sl slot holds the slack (>= 0) of |A'lam|^2 <= 1, no penetration slack, 0.5 a^2
        self.Ts, self.L = float(Ts), float(L)
        self.x0 = torch.tensor(np.asarray(x0, float).ravel())
        self.xF = torch.tensor(np.asarray(xF, float).ravel())
        self.vOb = [int(v) for v in np.asarray(vOb).ravel()]
        self.nOb = len(self.vOb)
        self.M = sum(self.vOb)
        self.A = torch.tensor(np.asarray(A, float).reshape(self.M, 2))
        self.b = torch.tensor(np.asarray(b, float).ravel())
        self.rx = torch.tensor(np.asarray(rx, float).ravel())
        self.ry = torch.tensor(np.asarray(ry, float).ravel())
        self.ryaw = torch.tensor(np.asarray(ryaw, float).ravel())
        self.fixTime = int(fixTime)
        ego = np.asarray(ego, float).ravel()
        W_ev, L_ev = ego[1] + ego[3], ego[0] + ego[2]
        self.g = torch.tensor([L_ev / 2, W_ev / 2, L_ev / 2, W_ev / 2])
        self.off = (ego[0] + ego[2]) / 2 - ego[2]
        self.XYb = np.asarray(XYb, float).ravel()
        N1, nOb, M = N + 1, self.nOb, self.M
        # flat layout
        o = 0
        self.ix = slice(o, o + 4 * N); o += 4 * N           # x_1..x_N (stage major); x_0 const
        self.it = o; o += 1                                   # t
        self.iu = slice(o, o + 2 * N); o += 2 * N
        self.il = slice(o, o + M * N1); o += M * N1           # lam[k, row]
        self.im = slice(o, o + 4 * nOb * N1); o += 4 * nOb * N1
        self.isl = slice(o, o + nOb * N1); o += nOb * N1
        self.iss = slice(o, o + N); o += N                    # steering-rate slack
        self.iso = slice(o, o + nOb * N1); o += nOb * N1      # obstacle-row slack
        self.n = o
        lb = -np.inf * np.ones(o); ub = np.inf * np.ones(o)
        xl = np.array([self.XYb[0], self.XYb[2], -np.inf, -1.0])
        xu = np.array([self.XYb[1], self.XYb[3], np.inf, 2.0])
        lb[self.ix] = np.tile(xl, N); ub[self.ix] = np.tile(xu, N)
        lb[self.it], ub[self.it] = 0.8, 1.2
        lb[self.iu] = np.tile([-0.6, -0.4], N); ub[self.iu] = np.tile([0.6, 0.4], N)
        lb[self.il] = 0; lb[self.im] = 0
        lb[self.iss] = -0.6; ub[self.iss] = 0.6
        lb[self.iso] = 0
        if self.dist:
            lb[self.isl] = 0
        self.lb, self.ub = lb, ub
        # multiplicity of the bound barrier (t is N+1 copies in the reference)
        self.mult = np.ones(o); self.mult[self.it] = N + 1
        if self.fixTime:
            lb[self.it] = ub[self.it] = 1.0
        self.m = 4 * N + 4 + N + 4 * nOb * N1

    def unpack(self, v):
        N, N1, nOb, M = self.N, self.N + 1, self.nOb, self.M
        x = torch.cat([self.x0[None, :], v[self.ix].reshape(N, 4)], 0)
        t = v[self.it] if not self.fixTime else torch.tensor(1.0)
        u = v[self.iu].reshape(N, 2)
        lam = v[self.il].reshape(N1, M)
        mu = v[self.im].reshape(N1, nOb, 4)
        sl = v[self.isl].reshape(N1, nOb)
        ss = v[self.iss]
        so = v[self.iso].reshape(N1, nOb)
        return x, t, u, lam, mu, sl, ss, so

    def f(self, v):
        x, t, u, lam, mu, sl, ss, so = self.unpack(v)
        N = self.N
        wa, wpsi = (0.5, 1e-2) if self.fixTime else (0.5 if self.dist else 0.1, 1e-4)
        w = torch.cat([torch.zeros(1, 2), u[:-1]], 0)
        q = t * self.Ts
        J = (0.01 * u[:, 0] ** 2 + wa * u[:, 1] ** 2).sum()
        J = J + 0.1 * (((u - w) / q) ** 2).sum()
        if not self.fixTime:
            J = J + (N + 1) * (0.5 * t + t ** 2)
        J = J + 1e-4 * (x[:, 3] ** 2).sum()
        J = J + (1e-3 * (x[:, 0] - self.rx) ** 2 + 1e-3 * (x[:, 1] - self.ry) ** 2
                 + wpsi * (x[:, 2] - self.ryaw) ** 2).sum()
        if not self.dist:
            J = J + (1e2 * sl + 1e4 * sl ** 2).sum()
        return J

    def c(self, v):
        x, t, u, lam, mu, sl, ss, so = self.unpack(v)
        N, Ts, L = self.N, self.Ts, self.L
        q = t * Ts
        X, Y, psi, vel = x[:-1, 0], x[:-1, 1], x[:-1, 2], x[:-1, 3]
        de, a = u[:, 0], u[:, 1]
        s = vel + q / 2 * a
        phi = psi + q / 2 * vel * torch.tan(de) / L
        Fx = torch.stack([X + q * s * torch.cos(phi), Y + q * s * torch.sin(phi),
                          psi + q * s * torch.tan(de) / L, vel + q * a], 1)
        cdyn = (x[1:] - Fx).reshape(-1)
        cterm = x[-1] - self.xF
        w0 = torch.cat([torch.zeros(1), u[:-1, 0]])
        csteer = (w0 - u[:, 0]) / q - ss
        cob = []
        r0 = 0
        cs, sn = torch.cos(x[:, 2]), torch.sin(x[:, 2])
        for j, vj in enumerate(self.vOb):
            Aj = self.A[r0:r0 + vj]; bj = self.b[r0:r0 + vj]; lj = lam[:, r0:r0 + vj]
            r0 += vj
            p = lj @ Aj                      # (N+1, 2)
            beta = lj @ bj
            m = mu[:, j]
            c1 = p[:, 0] ** 2 + p[:, 1] ** 2 - 1 + (sl[:, j] if self.dist else 0.0)
            c2 = m[:, 0] - m[:, 2] + cs * p[:, 0] + sn * p[:, 1]
            c3 = m[:, 1] - m[:, 3] - sn * p[:, 0] + cs * p[:, 1]
            c4 = (-(m * self.g).sum(1) + (x[:, 0] + cs * self.off) * p[:, 0]
                  + (x[:, 1] + sn * self.off) * p[:, 1] - beta + (0.0 if self.dist else sl[:, j]) - DMIN - so[:, j])
            cob.append(torch.stack([c1, c2, c3, c4], 1))
        cob = torch.stack(cob, 1).reshape(-1)    # [k, j, 4]
        return torch.cat([cdyn, cterm, csteer, cob])

    # numpy-facing derivative API -------------------------------------------------
    def eval_all(self, v, y):
        vt = torch.tensor(v, requires_grad=True)
        yt = torch.tensor(y)
        fval = self.f(vt)
        g, = torch.autograd.grad(fval, vt, create_graph=False)
        cval = self.c(vt.detach())
        Jm = torch.autograd.functional.jacobian(self.c, vt.detach())
        Lag = lambda w: self.f(w) + (yt * self.c(w)).sum()
        H = torch.autograd.functional.hessian(Lag, vt.detach())
        return fval.item(), g.numpy(), cval.numpy(), Jm.numpy(), H.numpy()

    def fc(self, v):
        vt = torch.tensor(v)
        return self.f(vt).item(), self.c(vt).numpy()

    def pack_start(self, xWS, uWS, lWS, nWS, t0=1.0):
        """xWS (N+1)x4, uWS Nx2, lWS (N+1)xM, nWS (N+1)x4nOb  (reference :213-222)."""
        v = np.zeros(self.n)
        v[self.ix] = np.asarray(xWS, float)[1:self.N + 1].reshape(-1)
        v[self.it] = t0
        v[self.iu] = np.asarray(uWS, float)[:self.N].reshape(-1)
        v[self.il] = np.asarray(lWS, float).reshape(-1)
        v[self.im] = np.asarray(nWS, float).reshape(-1)
        # slacks of the inequality rows start at the row value (then pushed inside)
        vt = torch.tensor(v)
        cv = self.c(vt).numpy()
        N, N1, nOb = self.N, self.N + 1, self.nOb
        o = 4 * N + 4
        v[self.iss] = cv[o:o + N]
        cob = cv[o + N:].reshape(N1, nOb, 4)
        v[self.iso] = cob[:, :, 3].reshape(-1)
        return v
