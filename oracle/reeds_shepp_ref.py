"""TEST INFRASTRUCTURE (checker only; nothing in obca_amd/ imports it).

CPU restatement of the path families of the reference's Reeds-Shepp module, AutonomousParking/reeds_shepp.jl:
    mod2pi :146-156, polar :139-143, LSL :159-169, LSR :172-187, LRL :190-204, set_path :207-230 (incl. its duplicate test, which is a signed
    sum, not a distance -- kept), SCS / SLS :233-269, CSC :272-315, CCC :318-368, calc_tauOmega :371-387, LRLRn / LRLRp :390-424, CCCC :427-478,
    LRSR / LRSL :481-516, CCSC :519-621, LRSLR :624-642, CCSCC :645-670, generate_path :782-800, calc_shortest_path_length :79-96,
    and the integration of a word into poses (interpolate :744-779) in closed form.
It returns what the reference's `calc_paths` / `calc_shortest_path` decide: the candidate words with their signed segment lengths and the shortest
one.  The reference cannot run in this image (Julia 0.6); the module is pinned by the reference's own acceptance criteria (check_path :846-869: every
candidate path ends on the goal pose) on its six fixed cases and on its random distribution (tests/test_planner_cpu.py).
"""
import math


def mod2pi(x):                                            # :146-156 (Julia mod: result in [0, 2 pi), then folded to (-pi, pi])
    v = math.fmod(x, 2.0 * math.pi)
    if v < 0:
        v += 2.0 * math.pi
    if v < -math.pi:
        v += 2.0 * math.pi
    elif v > math.pi:
        v -= 2.0 * math.pi
    return v


def polar(x, y):
    return math.hypot(x, y), math.atan2(y, x)


def LSL(x, y, phi):
    u, t = polar(x - math.sin(phi), y - 1.0 + math.cos(phi))
    if t >= 0.0:
        v = mod2pi(phi - t)
        if v >= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def LSR(x, y, phi):
    u1, t1 = polar(x + math.sin(phi), y - 1.0 - math.cos(phi))
    u1 = u1 ** 2
    if u1 >= 4.0:
        u = math.sqrt(u1 - 4.0)
        theta = math.atan2(2.0, u)
        t = mod2pi(t1 + theta)
        v = mod2pi(t - phi)
        if t >= 0.0 and v >= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def LRL(x, y, phi):
    u1, t1 = polar(x - math.sin(phi), y - 1.0 + math.cos(phi))
    if u1 <= 4.0:
        u = -2.0 * math.asin(0.25 * u1)
        t = mod2pi(t1 + 0.5 * u + math.pi)
        v = mod2pi(phi - t + u)
        if t >= 0.0 and u <= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def set_path(paths, lengths, ctypes):                     # :207-230
    for (l0, c0) in paths:
        if c0 == ctypes and sum(a - b for a, b in zip(l0, lengths)) <= 0.01:
            return paths                                  # "same path exists" (signed sum, as in the reference)
    if sum(abs(l) for l in lengths) >= 0.01:               # (the reference asserts this with Base.Test.@test)
        paths.append((list(lengths), list(ctypes)))
    return paths


def SLS(x, y, phi):
    phi = mod2pi(phi)
    if y > 0.0 and 0.0 < phi < math.pi * 0.99:
        xd = -y / math.tan(phi) + x
        return True, xd - math.tan(phi / 2.0), phi, math.sqrt((x - xd) ** 2 + y ** 2) - math.tan(phi / 2.0)
    if y < 0.0 and 0.0 < phi < math.pi * 0.99:
        xd = -y / math.tan(phi) + x
        return True, xd - math.tan(phi / 2.0), phi, -math.sqrt((x - xd) ** 2 + y ** 2) - math.tan(phi / 2.0)
    return False, 0.0, 0.0, 0.0


def SCS(x, y, phi, paths):
    f, t, u, v = SLS(x, y, phi)
    if f:
        set_path(paths, [t, u, v], ["S", "L", "S"])
    f, t, u, v = SLS(x, -y, -phi)
    if f:
        set_path(paths, [t, u, v], ["S", "R", "S"])
    return paths


def _four(fn, x, y, phi, paths, w, wm, build):
    """the reference's four reflections of a family: (x,y,phi), time flip (-x,y,-phi), reflection (x,-y,-phi), both (-x,-y,phi)"""
    for (xx, yy, pp, sgn, types) in ((x, y, phi, 1, w), (-x, y, -phi, -1, w), (x, -y, -phi, 1, wm), (-x, -y, phi, -1, wm)):
        f, t, u, v = fn(xx, yy, pp)
        if f:
            set_path(paths, [sgn * l for l in build(t, u, v)], list(types))


def CSC(x, y, phi, paths):
    _four(LSL, x, y, phi, paths, "LSL", "RSR", lambda t, u, v: [t, u, v])
    _four(LSR, x, y, phi, paths, "LSR", "RSL", lambda t, u, v: [t, u, v])
    return paths


def CCC(x, y, phi, paths):
    _four(LRL, x, y, phi, paths, "LRL", "RLR", lambda t, u, v: [t, u, v])
    xb = x * math.cos(phi) + y * math.sin(phi); yb = x * math.sin(phi) - y * math.cos(phi)          # backwards
    _four(LRL, xb, yb, phi, paths, "LRL", "RLR", lambda t, u, v: [v, u, t])
    return paths


def calc_tauOmega(u, v, xi, eta, phi):
    delta = mod2pi(u - v)
    A = math.sin(u) - math.sin(delta); B = math.cos(u) - math.cos(delta) - 1.0
    t1 = math.atan2(eta * A - xi * B, xi * A + eta * B)
    t2 = 2.0 * (math.cos(delta) - math.cos(v) - math.cos(u)) + 3.0
    tau = mod2pi(t1 + math.pi) if t2 < 0 else mod2pi(t1)
    return tau, mod2pi(tau - u + v - phi)


def LRLRn(x, y, phi):
    xi = x + math.sin(phi); eta = y - 1.0 - math.cos(phi)
    rho = 0.25 * (2.0 + math.sqrt(xi * xi + eta * eta))
    if rho <= 1.0:
        u = math.acos(rho)
        t, v = calc_tauOmega(u, -u, xi, eta, phi)
        if t >= 0.0 and v <= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def LRLRp(x, y, phi):
    xi = x + math.sin(phi); eta = y - 1.0 - math.cos(phi)
    rho = (20.0 - xi * xi - eta * eta) / 16.0
    if 0.0 <= rho <= 1.0:
        u = -math.acos(rho)
        if u >= -0.5 * math.pi:
            t, v = calc_tauOmega(u, u, xi, eta, phi)
            if t >= 0.0 and v >= 0.0:
                return True, t, u, v
    return False, 0.0, 0.0, 0.0


def CCCC(x, y, phi, paths):
    _four(LRLRn, x, y, phi, paths, "LRLR", "RLRL", lambda t, u, v: [t, u, -u, v])
    _four(LRLRp, x, y, phi, paths, "LRLR", "RLRL", lambda t, u, v: [t, u, u, v])
    return paths


def LRSR(x, y, phi):
    xi = x + math.sin(phi); eta = y - 1.0 - math.cos(phi)
    rho, theta = polar(-eta, xi)
    if rho >= 2.0:
        t = theta; u = 2.0 - rho; v = mod2pi(t + 0.5 * math.pi - phi)
        if t >= 0.0 and u <= 0.0 and v <= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def LRSL(x, y, phi):
    xi = x - math.sin(phi); eta = y - 1.0 + math.cos(phi)
    rho, theta = polar(xi, eta)
    if rho >= 2.0:
        r = math.sqrt(rho * rho - 4.0)
        u = 2.0 - r
        t = mod2pi(theta + math.atan2(r, -2.0))
        v = mod2pi(phi - 0.5 * math.pi - t)
        if t >= 0.0 and u <= 0.0 and v <= 0.0:
            return True, t, u, v
    return False, 0.0, 0.0, 0.0


def CCSC(x, y, phi, paths):
    h = 0.5 * math.pi
    _four(LRSL, x, y, phi, paths, "LRSL", "RLSR", lambda t, u, v: [t, -h, u, v])
    _four(LRSR, x, y, phi, paths, "LRSR", "RLSL", lambda t, u, v: [t, -h, u, v])
    xb = x * math.cos(phi) + y * math.sin(phi); yb = x * math.sin(phi) - y * math.cos(phi)          # backwards
    _four(LRSL, xb, yb, phi, paths, "LSRL", "RSLR", lambda t, u, v: [v, u, -h, t])
    _four(LRSR, xb, yb, phi, paths, "RSRL", "LSLR", lambda t, u, v: [v, u, -h, t])
    return paths


def LRSLR(x, y, phi):
    xi = x + math.sin(phi); eta = y - 1.0 - math.cos(phi)
    rho, theta = polar(xi, eta)
    if rho >= 2.0:
        u = 4.0 - math.sqrt(rho * rho - 4.0)
        if u <= 0.0:
            t = mod2pi(math.atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta))
            v = mod2pi(t - phi)
            if t >= 0.0 and v >= 0.0:
                return True, t, u, v
    return False, 0.0, 0.0, 0.0


def CCSCC(x, y, phi, paths):
    h = 0.5 * math.pi
    _four(LRSLR, x, y, phi, paths, "LRSLR", "RLSRL", lambda t, u, v: [t, -h, u, -h, v])
    return paths


def generate_path(q0, q1, maxc):                          # :782-800: candidate words in the start frame, lengths in units of the turning radius
    dx = q1[0] - q0[0]; dy = q1[1] - q0[1]; dth = q1[2] - q0[2]
    c = math.cos(q0[2]); s = math.sin(q0[2])
    x = (c * dx + s * dy) * maxc; y = (-s * dx + c * dy) * maxc
    paths = []
    for fam in (SCS, CSC, CCC, CCCC, CCSC, CCSCC):
        fam(x, y, dth, paths)
    return paths


def end_pose(q0, lengths, ctypes, maxc):
    """closed-form integration of a word from q0 (what interpolate :744-779 samples): lengths in units of 1 / maxc, signed"""
    x, y, th = q0
    for l, m in zip(lengths, ctypes):
        if m == "S":
            x += l / maxc * math.cos(th); y += l / maxc * math.sin(th)
        else:
            sg = 1.0 if m == "L" else -1.0
            x += (math.sin(th + sg * l) - math.sin(th)) * sg / maxc
            y += (-math.cos(th + sg * l) + math.cos(th)) * sg / maxc
            th += sg * l
    return x, y, th


def calc_paths(q0, q1, maxc):
    """[(total length in metres, signed segment lengths in metres, word)] of every candidate (calc_paths :99-120)"""
    return [(sum(abs(l) for l in ls) / maxc, [l / maxc for l in ls], "".join(ct)) for ls, ct in generate_path(q0, q1, maxc)]


def shortest(q0, q1, maxc):
    """the candidate calc_shortest_path :59-76 returns (the LAST of equal minima: its comparison is <=)"""
    best = None
    for p in calc_paths(q0, q1, maxc):
        if best is None or p[0] <= best[0]:
            best = p
    return best
