"""Which interpolation points for Winograd F(4x4, 3x3) in fp32?  (numpy model of the three stages, CPU only)

Cook-Toom matrices for arbitrary points are generated in exact rational arithmetic, checked against a direct 1-D
correlation, and the fp32 error of a 256-channel 3x3 layer (leaky-ReLU'd unit-variance input, He-scaled filters, filters
transformed in fp64 and rounded once - as rib_finalize_weights does) is measured against the fp64 direct convolution.

    python tools/probes/wino_points.py

Round-2 result (max / rms error on outputs of magnitude ~4):
    direct fp32                      1.4e-06
    F(2x2)  0, +-1                   2.7e-06 / 4.6e-07
    F(4x4)  0, +-1, +-2  (textbook)  4.9e-05 / 2.9e-06
    F(4x4)  0, +-1, +-1/2            3.5e-05 / 2.9e-06
    F(4x4)  0, +-1/2, +-3/2          1.8e-05 / 1.6e-06
    F(4x4)  0, +-3/4, +-3/2          9.3e-06 / 1.4e-06     <- used (kernels.hip.h k_wino4_in / k_wino4_out): B^T, A^T dyadic
"""
from fractions import Fraction as Fr

import numpy as np


def polymul(a, b):
    r = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            r[i + j] += x * y
    return r


def cook_toom(m, r, pts):
    """A^T [m, n], G [n, r], B^T [n, n] (n = m + r - 1) for the finite points `pts` (n - 1 of them) plus infinity."""
    n = m + r - 1
    a = [Fr(p) for p in pts]
    assert len(a) == n - 1
    f = []
    for j in range(n - 1):
        p = [Fr(1)]
        for k in range(n - 1):
            if k != j:
                p = polymul(p, [-a[k], Fr(1)])
        f.append(p)
    full = [Fr(1)]
    for k in range(n - 1):
        full = polymul(full, [-a[k], Fr(1)])
    N = [sum(c * a[j] ** i for i, c in enumerate(f[j])) for j in range(n - 1)]
    AT = [[a[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = [[a[j] ** k / N[j] for k in range(r)] for j in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    BT = [f[j] + [Fr(0)] for j in range(n - 1)] + [full]
    return AT, G, BT


def to_f(M):
    return np.array([[float(x) for x in row] for row in M])


def check_1d(AT, G, BT, m, r):
    rng = np.random.default_rng(1)
    d = rng.standard_normal(m + r - 1)
    g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()


def main():
    rng = np.random.default_rng(0)
    C, K, H = 256, 32, 24
    x = rng.standard_normal((C, H + 2, H + 2)).astype(np.float32)
    x = np.where(x > 0, x, 0.2 * x)
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2 / (C * 9))).astype(np.float32)

    def direct(dt):
        xx, ww = x.astype(dt), w.astype(dt)
        out = np.zeros((K, H, H), dt)
        for r in range(3):
            for s in range(3):
                out += np.einsum("kc,chw->khw", ww[:, :, r, s], xx[:, r:r + H, s:s + H])
        return out

    ref = direct(np.float64)
    print("%-26s max %.2e" % ("direct fp32", np.abs(direct(np.float32) - ref).max()))

    def wino(AT, G, BT, m, dt=np.float32):
        a = m + 2
        U = np.einsum("ir,kcrs,js->ijkc", G, w.astype(np.float64), G).astype(dt)
        BTf, ATf = BT.astype(dt), AT.astype(dt)
        ty = np.arange(0, H, m)
        d = np.stack([np.stack([x[:, y:y + a, xx:xx + a] for xx in ty]) for y in ty])
        V = np.einsum("ir,yxcrs->yxcis", BTf, d).astype(dt)
        V = np.einsum("js,yxcis->yxcij", BTf, V).astype(dt)
        M = np.einsum("ijkc,yxcij->yxkij", U, V).astype(dt)
        Y = np.einsum("ri,yxkij->yxkrj", ATf, M).astype(dt)
        Y = np.einsum("sj,yxkrj->yxkrs", ATf, Y).astype(dt)
        out = np.zeros((K, H, H), dt)
        for i, y in enumerate(ty):
            for j, xx in enumerate(ty):
                out[:, y:y + m, xx:xx + m] = Y[i, j]
        return out

    h = Fr(1, 2)
    sets = {
        "F(2x2) 0,+-1": [0, 1, -1],
        "F(4x4) 0,+-1,+-2": [0, 1, -1, 2, -2],
        "F(4x4) 0,+-1,+-1/2": [0, 1, -1, h, -h],
        "F(4x4) 0,+-1/2,+-3/2": [0, h, -h, 3 * h, -3 * h],
        "F(4x4) 0,+-3/4,+-3/2": [0, Fr(3, 4), -Fr(3, 4), Fr(3, 2), -Fr(3, 2)],
        "F(4x4) 0,+-5/8,+-5/4": [0, Fr(5, 8), -Fr(5, 8), Fr(5, 4), -Fr(5, 4)],
        "F(4x4) 0,+-1/2,+-2": [0, h, -h, 2, -2],
    }
    for name, pts in sets.items():
        m = len(pts) - 1
        ATq, Gq, BTq = cook_toom(m, 3, pts)
        AT, G, BT = to_f(ATq), to_f(Gq), to_f(BTq)
        assert check_1d(AT, G, BT, m, 3) < 1e-12
        e = np.abs(wino(AT, G, BT, m) - ref)
        print("%-26s max %.2e rms %.2e" % (name, e.max(), np.sqrt((e ** 2).mean())))
    ATq, Gq, BTq = cook_toom(4, 3, sets["F(4x4) 0,+-3/4,+-3/2"])
    for nm, M in (("A^T", ATq), ("G", Gq), ("B^T", BTq)):
        print(nm)
        for row in M:
            print("   ", [str(v) for v in row])


if __name__ == "__main__":
    main()
