"""Accuracy of F(4x4,3x3) Winograd convolution in fp32 for different interpolation point sets (CPU, numpy).

Cook-Toom construction with exact fractions, then a simulated fp32 pipeline (transforms + channel-summed products in fp32)
against the fp64 direct convolution.  python tools/experiments/winograd_points.py"""
from fractions import Fraction as Fr
import itertools
import numpy as np


def cook_toom(points, m=4, r=3):
    """points: n-1 finite points (n = m + r - 1); the n-th is infinity.  Returns AT (m x n), G (n x r), BT (n x n) as Fractions,
    scaled as wincnn does (Lagrange denominators in G)."""
    n = m + r - 1
    p = [Fr(x) for x in points]
    AT = [[(p[j] ** i if j < n - 1 else Fr(1 if i == m - 1 else 0)) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fr(1)
        for k in range(n - 1):
            if k != j:
                f *= (p[j] - p[k])
        G.append([p[j] ** k / f for k in range(r)])
    G.append([Fr(1 if k == r - 1 else 0) for k in range(r)])
    # solve for BT: sum_j AT[i][j] G[j][k] BT[j][l] = [l == i + k]
    rows, rhs = [], []
    for i in range(m):
        for k in range(r):
            rows.append([AT[i][j] * G[j][k] for j in range(n)])
            rhs.append([Fr(1 if l == i + k else 0) for l in range(n)])
    BT = solve(rows, rhs, n)
    return AT, G, BT


def solve(rows, rhs, n):
    """Least-squares-free exact solve of an overdetermined consistent system by Gaussian elimination on [rows | rhs]."""
    M = [list(a) + list(b) for a, b in zip(rows, rhs)]
    piv = 0
    for c in range(n):
        pr = next(i for i in range(piv, len(M)) if M[i][c] != 0)
        M[piv], M[pr] = M[pr], M[piv]
        inv = 1 / M[piv][c]
        M[piv] = [x * inv for x in M[piv]]
        for i in range(len(M)):
            if i != piv and M[i][c] != 0:
                f = M[i][c]
                M[i] = [x - f * y for x, y in zip(M[i], M[piv])]
        piv += 1
    for i in range(n, len(M)):
        assert all(x == 0 for x in M[i]), "inconsistent"
    return [M[j][n:] for j in range(n)]


def f32(mat):
    return np.array([[float(x) for x in row] for row in mat], dtype=np.float32)


def simulate(points, C=128, T=256, K=32, seed=0, relu=True):
    AT, G, BT = cook_toom(points)
    at, g, bt = f32(AT), f32(G), f32(BT)
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((T, C, 6, 6)).astype(np.float32)
    if relu:
        d = np.maximum(d, 0)
    w = (rng.standard_normal((K, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).astype(np.float32)
    # fp64 direct
    d64, w64 = d.astype(np.float64), w.astype(np.float64)
    ref = np.zeros((T, K, 4, 4))
    for a in range(3):
        for b in range(3):
            ref += np.einsum("tcij,kc->tkij", d64[:, :, a:a + 4, b:b + 4], w64[:, :, a, b])
    # fp32 winograd: every matmul in fp32 (numpy float32 einsum accumulates in fp32)
    V = np.einsum("ia,tcab,jb->tcij", bt, d, bt, optimize=False).astype(np.float32)
    U = np.einsum("ia,kcab,jb->kcij", g, w, g, optimize=False).astype(np.float32)
    M = np.einsum("tcij,kcij->tkij", V, U, optimize=False).astype(np.float32)
    Y = np.einsum("ia,tkab,jb->tkij", at, M, at, optimize=False).astype(np.float32)
    err = Y.astype(np.float64) - ref
    # fp32 direct for the yardstick
    dd = np.zeros((T, K, 4, 4), dtype=np.float32)
    for a in range(3):
        for b in range(3):
            dd += np.einsum("tcij,kc->tkij", d[:, :, a:a + 4, b:b + 4], w[:, :, a, b]).astype(np.float32)
    errd = dd.astype(np.float64) - ref
    nrm = np.sqrt((ref ** 2).mean())
    return np.sqrt((err ** 2).mean()) / nrm, np.sqrt((errd ** 2).mean()) / nrm


if __name__ == "__main__":
    H = Fr(1, 2)
    cands = {
        "0,1,-1,2,-2 (Lavin, current)": [0, 1, -1, 2, -2],
        "0,1,-1,1/2,-1/2": [0, 1, -1, H, -H],
        "0,1,-1,1/2,-2": [0, 1, -1, H, -2],
        "0,1,-1,2,-1/2": [0, 1, -1, 2, -H],
        "0,1,-1,1/2,2": [0, 1, -1, H, 2],
        "0,1/2,-1/2,1,-1 (same set)": [0, H, -H, 1, -1],
        "0,1,-1,3/2,-3/2": [0, 1, -1, Fr(3, 2), Fr(-3, 2)],
        "0,1/2,-1/2,3/2,-3/2": [0, H, -H, Fr(3, 2), Fr(-3, 2)],
        "0,1,-1,2/3,-2/3(?)": [0, 1, -1, Fr(2, 3), Fr(-2, 3)],
        "0,3/4,-3/4,3/2,-3/2": [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)],
    }
    for name, pts in cands.items():
        e = [simulate(pts, seed=s) for s in range(2)]
        print("%-32s winograd fp32 rel rms err %.3e   direct fp32 %.3e   ratio %.1f" % (
            name, np.mean([x[0] for x in e]), np.mean([x[1] for x in e]), np.mean([x[0] for x in e]) / np.mean([x[1] for x in e])))


def scan():
    vals = [Fr(1, 4), Fr(1, 3), Fr(3, 8), Fr(1, 2), Fr(5, 8), Fr(2, 3), Fr(3, 4), Fr(7, 8), Fr(1), Fr(5, 4), Fr(3, 2), Fr(7, 4), Fr(2), Fr(5, 2), Fr(3)]
    res = []
    for a, b in itertools.combinations(vals, 2):
        e = simulate([0, a, -a, b, -b], C=64, T=64, K=16)
        res.append((e[0] / e[1], str(a), str(b)))
    res.sort()
    for r in res[:12]:
        print("0, +-%s, +-%s: ratio to direct %.1f" % (r[1], r[2], r[0]))
    print("worst:", res[-1])
