"""
TEST INFRASTRUCTURE ONLY (oracle).  The reference's quadcopter NLP exactly as JuMP 0.18 hands it to IPOPT (QuadcopterNavigation/QuadcopterSignedDist.jl:34-197) -- NOT the
reformulated problem the kernels and the C oracle solve -- for the dense Algorithm A of oracle/ipm_ref80.py (`attempt`), at the benchmark size N = 60: 3 168 variables,
1 109 equality rows, 3 473 inequality rows with an IPOPT slack each, an 11 223-dimensional KKT system.

What is deliberately NOT shared with obca_amd/csrc/obca_quad_*.h and oracle/obca_oracle_quad.c:
  * timeScale is the vector of N + 1 variables tied by `timeScale[i] == timeScale[i+1]` (:152); stage i uses timeScale[i] (:130-150);
  * x[:, 1] == x0 is kept as twelve equality rows (:107-119), x[:, 1] is a variable -- so the single-index terms x[10], x[11], x[12] of the rate rows (:148-150, SURVEY Q2:
    Julia's linear index = the stage-1 rates) are variables with derivatives, not constants;
  * every bound is a constraint ROW with a slack (`@constraint(m, lb <= x <= ub)`, `l .>= 0`, `slack .>= 0`), the separation rows `>= R` are inequality rows;
  * IPOPT's gradient-based scaling (nlp_scaling_max_gradient = 100) is applied to the objective and the rows at the starting point -- the objective factor is ~0.05 here, which the
    kernels do not apply: the dense solve terminates at a looser effective tolerance than the kernels;
  * derivatives by torch autograd, one dense Bunch-Kaufman LDL' of the full system per inertia trial, inertia counted.
Options of the reference's call (:28-31): tol 1e-5, no max_iter (IPOPT: 3 000), min_hessian_perturbation 1e-10, jacobian_regularization_value 1e-7, alpha_for_y = min,
recalc_y = "no"; IPOPT's defaults max_soc = 4 and least-squares initial multipliers.
"""
import numpy as np
import torch
from ipm_ref80 import RefNLP, Opts, attempt

torch.set_default_dtype(torch.float64)
MASS, GRAV, KF, KM, ARM = 0.5, 9.81, 0.0611, 0.0015, 0.225
INERT = (3.9e-3, 4.4e-3, 4.9e-3)
XLB = np.array([0, 0, 0, -3, -0.2, -0.2, -1, -1, -1, -1, -1, -1.0])
XUB = np.array([10, 10, 5, 3, 0.2, 0.2, 1, 1, 1, 1, 1, 1.0])


class QuadOpts(Opts):
    max_iter = 3000; dw_min = 1e-10; recalc_y = False


class RefQuadNLP:
    """variable order = JuMP's declaration order (:34-48): x (12, N+1), timeScale (N+1), u (4, N), l1..l5 (6, N+1) each, slack (5, N+1), column-major"""
    derivs = RefNLP.derivs

    def __init__(self, x0, xF, N, Ts, R, ob):
        self.N = N; N1 = N + 1
        self.Ts, self.R = float(Ts), float(R)
        self.x0 = torch.tensor(np.asarray(x0, float).ravel()); self.xF = torch.tensor(np.asarray(xF, float).ravel())
        self.ob = torch.tensor(np.asarray(ob, float).reshape(5, 6))              # b of A = [I; -I]: [xmax, ymax, zmax, -xmin, -ymin, -zmin] per box (:162-166)
        self.wH = float(np.sqrt(MASS * GRAV / (KF * 4)))
        o = 0
        self.ix = o; o += 12 * N1; self.it = o; o += N1; self.iu = o; o += 4 * N; self.il = o; o += 30 * N1; self.isl = o; o += 5 * N1
        self.n = o
        self.mc = 12 + 12 + 12 * N + N + 5 * N1
        self.md = 4 * N + 12 * N1 + N1 + 30 * N1 + 5 * N1 + 5 * N1
        dL = [1.2] * (4 * N) + list(np.tile(XLB, N1)) + [0.5] * N1 + [0.0] * (30 * N1 + 5 * N1) + [self.R] * (5 * N1)          # :82-105, :165-193
        dU = [7.8] * (4 * N) + list(np.tile(XUB, N1)) + [2.0] * N1 + [np.inf] * (30 * N1 + 5 * N1 + 5 * N1)
        self.dL = np.array(dL, float); self.dU = np.array(dU, float)
        assert len(dL) == self.md == len(dU)

    def split(self, v):
        N, N1 = self.N, self.N + 1
        x = v[self.ix:self.it].reshape(N1, 12); ts = v[self.it:self.iu]; u = v[self.iu:self.il].reshape(N, 4)
        lam = v[self.il:self.isl].reshape(5, N1, 6)                              # l1 .. l5, each (6, N+1) column-major
        s = v[self.isl:].reshape(N1, 5)
        return x, ts, u, lam, s

    def f(self, v):      # :66-73
        x, ts, u, lam, s = self.split(v)
        J = 1e-3 * ((self.wH - u) ** 2).sum() + 1e-2 * ((u[:-1] - u[1:]) ** 2).sum() + 1e-4 * (x[:, 9:12] ** 2).sum()
        return J + (0.25 * ts + 5 * ts ** 2).sum() + (1e2 * s + 1e3 * s ** 2).sum() + 1e-4 * (lam ** 2).sum()

    def c(self, v):
        x, ts, u, lam, s = self.split(v); N = self.N
        X = x[:-1]; tau = (ts[:N] * self.Ts)[:, None]
        s4, c4, s5, c5, s6, c6 = torch.sin(X[:, 3]), torch.cos(X[:, 3]), torch.sin(X[:, 4]), torch.cos(X[:, 4]), torch.sin(X[:, 5]), torch.cos(X[:, 5])
        T4, S4 = s4 / c4, 1 / c4
        U = (u ** 2).sum(1); g = x[0, 9:12]                                     # x[10], x[11], x[12]: linear indices = stage 1 (:148-150)
        one = torch.ones(N, dtype=v.dtype)
        G = torch.stack([
            X[:, 6], X[:, 7], X[:, 8],
            c5 * X[:, 9] + s5 * X[:, 11],
            s5 * T4 * X[:, 9] + X[:, 10] - c5 * T4 * X[:, 11],
            -s5 * S4 * X[:, 9] + c5 * S4 * X[:, 11],
            KF / MASS * U * (s4 * c5 * s6 + s5 * c6),
            KF / MASS * U * (-s4 * c5 * c6 + s5 * s6),
            (KF * U * c4 * c5 - MASS * GRAV) / MASS,
            (ARM * KF * (u[:, 1] ** 2 - u[:, 3] ** 2) - (INERT[2] - INERT[1]) * g[1] * g[2] * one) / INERT[0],
            (ARM * KF * (u[:, 2] ** 2 - u[:, 0] ** 2) - (INERT[0] - INERT[2]) * g[0] * g[2] * one) / INERT[1],
            (KM * (u[:, 0] ** 2 - u[:, 1] ** 2 + u[:, 2] ** 2 - u[:, 3] ** 2) - (INERT[1] - INERT[0]) * g[0] * g[1] * one) / INERT[2]], 1)
        q = lam[:, :, :3] - lam[:, :, 3:]                                        # A' l  with A = [I; -I]
        return torch.cat([x[0] - self.x0, x[-1] - self.xF, (x[1:] - X - tau * G).reshape(-1), ts[:-1] - ts[1:], ((q ** 2).sum(2) - 1.0).reshape(-1)])

    def d(self, v):
        x, ts, u, lam, s = self.split(v)
        q = lam[:, :, :3] - lam[:, :, 3:]
        sep = -(lam * self.ob[:, None, :]).sum(2) + (x[None, :, :3] * q).sum(2) + 0.01 * s.T          # (5, N+1), :165-193
        return torch.cat([u.reshape(-1), x.reshape(-1), ts, lam.reshape(-1), s.reshape(-1), sep.reshape(-1)])

    def start(self, xWS, timeWS, lam0):
        """:198-210: timeScale <- timeWS, x <- xWS, u <- hover speed, slack <- 1; l <- lam0 (5, N+1, 6): the reference's 0.05 or the closed-form distance duals the library starts from"""
        v = np.zeros(self.n); N, N1 = self.N, self.N + 1
        v[self.ix:self.it] = np.asarray(xWS, float)[:N1].reshape(-1); v[self.it:self.iu] = float(timeWS); v[self.iu:self.il] = self.wH
        v[self.il:self.isl] = np.asarray(lam0, float).reshape(-1); v[self.isl:] = 1.0
        return v


def box_duals(xWS, ob):
    """closed-form dual solution of the point-to-box distance at every warm-start position (what the library's dual_ws = 1 start is; oracle/obca_oracle_quad.c: dual_ws_block)"""
    xWS = np.asarray(xWS, float); ob = np.asarray(ob, float).reshape(5, 6); N1 = len(xWS)
    lam = np.zeros((5, N1, 6))
    for j in range(5):
        hi, lo = ob[j, :3], -ob[j, 3:]
        for k in range(N1):
            p = xWS[k, :3]; d = p - np.clip(p, lo, hi); n2 = (d ** 2).sum(); q = np.zeros(3)
            if n2 > 1e-16:
                q = d / np.sqrt(n2)
            else:
                dh, dl = hi - p, p - lo; best, bd, sg = 0, 1e300, 1.0
                for i in range(3):
                    if dh[i] < bd: bd, best, sg = dh[i], i, 1.0
                    if dl[i] < bd: bd, best, sg = dl[i], i, -1.0
                q[best] = sg
            lam[j, k, :3] = np.maximum(q, 0); lam[j, k, 3:] = np.maximum(-q, 0)
    return lam
