"""
TEST INFRASTRUCTURE ONLY.  Flat torch-fp64 restatement of the quadcopter signed-distance NLP
(/root/reference/QuadcopterNavigation/QuadcopterSignedDist.jl:34-197) with autograd derivatives; used to cross-check the
closed-form derivatives and the structured Newton solve of oracle/obca_oracle_quad.c.  Same reformulations as the C oracle
(one scalar t with multiplicity N+1, x_0 eliminated, stage-1 gyroscopic constants, slack on the >= rows).
"""
import numpy as np
import torch

torch.set_default_dtype(torch.float64)
MASS, GRAV, KF, KM, ARM = 0.5, 9.81, 0.0611, 0.0015, 0.225
INERT = (3.9e-3, 4.4e-3, 4.9e-3)
XLB = np.array([0, 0, 0, -3, -0.2, -0.2, -1, -1, -1, -1, -1, -1.0])
XUB = np.array([10, 10, 5, 3, 0.2, 0.2, 1, 1, 1, 1, 1, 1.0])


class QuadNLP:
    def __init__(self, x0, xF, N, Ts, R, ob, dist=False):
        self.N, self.Ts, self.R = N, float(Ts), float(R)
        self.sw = 0.0 if dist else 1.0           # QuadcopterDist: the slack variable does not exist (kept in the vector, weight 0, no bound)
        self.x0 = torch.tensor(np.asarray(x0, float).ravel()); self.xF = torch.tensor(np.asarray(xF, float).ravel())
        self.ob = torch.tensor(np.asarray(ob, float).reshape(5, 6))
        self.wH = float(np.sqrt(MASS * GRAV / (KF * 4)))
        N1 = N + 1; o = 0
        self.ix = slice(o, o + 12 * N); o += 12 * N
        self.iu = slice(o, o + 4 * N); o += 4 * N
        self.it = o; o += 1
        self.il = slice(o, o + 30 * N1); o += 30 * N1
        self.isl = slice(o, o + 5 * N1); o += 5 * N1
        self.iso = slice(o, o + 5 * N1); o += 5 * N1
        self.n = o
        lb = -np.inf * np.ones(o); ub = np.inf * np.ones(o)
        lb[self.ix] = np.tile(XLB, N); ub[self.ix] = np.tile(XUB, N)
        lb[self.iu] = 1.2; ub[self.iu] = 7.8
        lb[self.it], ub[self.it] = 0.5, 2.0
        lb[self.il] = 0; lb[self.isl] = -np.inf if dist else 0; lb[self.iso] = 0
        if dist:
            lb[self.ix][9::12] = -1.5; ub[self.ix][9::12] = 3.0
        self.lb, self.ub = lb, ub
        self.mult = np.ones(o); self.mult[self.it] = N + 1
        self.m = 12 * N + 12 + 10 * N1

    def unpack(self, v):
        N, N1 = self.N, self.N + 1
        x = torch.cat([self.x0[None, :], v[self.ix].reshape(N, 12)], 0)
        return x, v[self.iu].reshape(N, 4), v[self.it], v[self.il].reshape(N1, 5, 6), v[self.isl].reshape(N1, 5), v[self.iso].reshape(N1, 5)

    def f(self, v):
        x, u, t, lam, s, so = self.unpack(v)
        J = 1e-3 * ((self.wH - u) ** 2).sum() + 1e-2 * ((u[:-1] - u[1:]) ** 2).sum() + 1e-4 * (x[:, 9:12] ** 2).sum()
        J = J + (self.N + 1) * (0.25 * t + 5 * t ** 2) + self.sw * (1e2 * s + 1e3 * s ** 2).sum() + 1e-4 * (lam ** 2).sum()
        return J

    def c(self, v):
        x, u, t, lam, s, so = self.unpack(v)
        X = x[:-1]; tau = t * self.Ts
        s4, c4, s5, c5, s6, c6 = torch.sin(X[:, 3]), torch.cos(X[:, 3]), torch.sin(X[:, 4]), torch.cos(X[:, 4]), torch.sin(X[:, 5]), torch.cos(X[:, 5])
        T4, S4 = s4 / c4, 1 / c4
        U = (u ** 2).sum(1); g = self.x0[9:12]
        G = torch.stack([
            X[:, 6], X[:, 7], X[:, 8],
            c5 * X[:, 9] + s5 * X[:, 11],
            s5 * T4 * X[:, 9] + X[:, 10] - c5 * T4 * X[:, 11],
            -s5 * S4 * X[:, 9] + c5 * S4 * X[:, 11],
            KF / MASS * U * (s4 * c5 * s6 + s5 * c6),
            KF / MASS * U * (-s4 * c5 * c6 + s5 * s6),
            (KF * U * c4 * c5 - MASS * GRAV) / MASS,
            (ARM * KF * (u[:, 1] ** 2 - u[:, 3] ** 2) - (INERT[2] - INERT[1]) * g[1] * g[2]) / INERT[0],
            (ARM * KF * (u[:, 2] ** 2 - u[:, 0] ** 2) - (INERT[0] - INERT[2]) * g[0] * g[2]) / INERT[1],
            (KM * (u[:, 0] ** 2 - u[:, 1] ** 2 + u[:, 2] ** 2 - u[:, 3] ** 2) - (INERT[1] - INERT[0]) * g[0] * g[1]) / INERT[2]], 1)
        cdyn = (x[1:] - X - tau * G).reshape(-1)
        cterm = x[-1] - self.xF
        q = lam[:, :, :3] - lam[:, :, 3:]
        c1 = (q ** 2).sum(2) - 1
        c2 = -(lam * self.ob[None]).sum(2) + (x[:, None, :3] * q).sum(2) + self.sw * 0.01 * s - self.R - so
        cob = torch.stack([c1, c2], 2).reshape(-1)
        return torch.cat([cdyn, cterm, cob])

    def eval_all(self, v, y):
        vt = torch.tensor(v); yt = torch.tensor(y)
        g = torch.autograd.functional.jacobian(self.f, vt)
        J = torch.autograd.functional.jacobian(self.c, vt)
        H = torch.autograd.functional.hessian(lambda w: self.f(w) + (yt * self.c(w)).sum(), vt)
        return self.f(vt).item(), g.numpy(), self.c(vt).numpy(), J.numpy(), H.numpy()
