"""fp32 error of Winograd F(m x m, 3 x 3) against the float64 convolution for candidate point sets (CPU, numpy).
Transforms in fp32, U = G g G^T rounded once from float64, channel sum in fp32 -- what the HIP kernels do."""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np


def cook_toom(points, m, r):
    """(AT m x n, G n x r, BT n x n) as float64 for finite `points` (n - 1 of them) + infinity."""
    n = m + r - 1
    assert len(points) == n - 1
    p = [Fr(x) for x in points]

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    AT = [[p[i] ** j for i in range(n - 1)] + [Fr(1) if j == m - 1 else Fr(0)] for j in range(m)]
    G, BT = [], []
    for i in range(n - 1):
        N = Fr(1)
        poly = [Fr(1)]
        for j in range(n - 1):
            if j != i:
                N *= p[i] - p[j]
                poly = polymul(poly, [-p[j], Fr(1)])
        G.append([p[i] ** k / N for k in range(r)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for j in range(n - 1):
        poly = polymul(poly, [-p[j], Fr(1)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def rescale_pow2(G, BT):
    """Move powers of two between rows of BT and G so that BT rows have max-abs in [1, 2): exponent-only, no rounding."""
    s = 2.0 ** np.floor(np.log2(np.abs(BT).max(axis=1)))
    return G * s[:, None], BT / s[:, None]


def run(points, m, C, N, seed=0, tiles=64, act='relu'):
    r = 3
    n = m + r - 1
    AT, G, BT = cook_toom(points, m, r)
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((tiles, n, n, C))
    if act == 'relu':
        d = np.maximum(d, 0)
    lim = np.sqrt(6.0 / (9 * C + 9 * N))               # glorot uniform, Keras default
    g = rng.uniform(-lim, lim, (3, 3, C, N))
    d32, g32 = d.astype(np.float32), g.astype(np.float32)
    # float64 reference on the fp32 inputs
    ref = np.zeros((tiles, m, m, N))
    for a in range(m):
        for b in range(m):
            ref[:, a, b] = np.einsum('tuvc,uvcn->tn', d32[:, a:a + 3, b:b + 3].astype(np.float64), g32.astype(np.float64))
    # exactness of the algorithm in float64
    U = np.einsum('au,bv,uvcn->abcn', G, G, g32.astype(np.float64))
    V = np.einsum('au,bv,tuvc->tabc', BT, BT, d32.astype(np.float64))
    Y = np.einsum('ia,jb,tabn->tijn', AT, AT, np.einsum('tabc,abcn->tabn', V, U))
    assert np.abs(Y - ref).max() < 1e-9 * max(1, np.abs(ref).max()), 'algorithm wrong'
    # fp32 emulation
    U32 = U.astype(np.float32)
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    V32 = np.einsum('au,tuvc->tavc', BT32, d32).astype(np.float32)
    V32 = np.einsum('bv,tavc->tabc', BT32, V32).astype(np.float32)
    M32 = np.zeros((tiles, n, n, N), dtype=np.float32)
    for a in range(n):
        for b in range(n):
            M32[:, a, b] = V32[:, a, b] @ U32[a, b]              # sgemm: fp32 accumulate
    Y32 = np.einsum('ia,tabn->tibn', AT32, M32).astype(np.float32)
    Y32 = np.einsum('jb,tibn->tijn', AT32, Y32).astype(np.float32)
    # direct fp32
    D32 = np.zeros((tiles, m, m, N), dtype=np.float32)
    for a in range(m):
        for b in range(m):
            D32[:, a, b] = d32[:, a:a + 3, b:b + 3].reshape(tiles, -1) @ g32.reshape(-1, N)
    scale = np.abs(ref).max()
    e_w = np.abs(Y32 - ref)
    e_d = np.abs(D32 - ref)
    return scale, e_w.max(), np.sqrt((e_w ** 2).mean()), e_d.max(), np.sqrt((e_d ** 2).mean())


if __name__ == '__main__':
    C, N = 256, 64
    print(f'C = {C}: max|out| scale, winograd max / rms error, direct fp32 max / rms')
    cases = [('F(2x2) {0,1,-1}', (0, 1, -1), 2),
             ('F(4x4) {0,1,-1,2,-2}', (0, 1, -1, 2, -2), 4),
             ('F(4x4) {0,1,-1,1/2,-1/2}', (0, 1, -1, Fr(1, 2), Fr(-1, 2)), 4),
             ('F(4x4) {0,1,-1,1/2,-2}', (0, 1, -1, Fr(1, 2), -2), 4),
             ('F(4x4) {0,1,-1,2,-1/2}', (0, 1, -1, 2, Fr(-1, 2)), 4),
             ('F(4x4) {0,1/2,-1/2,3/2,-3/2}', (0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)), 4),
             ('F(4x4) {0,3/4,-3/4,3/2,-3/2}', (0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)), 4),
             ('F(4x4) {0,1/2,-1/2,1,-1} = above', (0, Fr(1, 2), Fr(-1, 2), 1, -1), 4),
             ('F(4x4) {0,5/8,-5/8,5/4,-5/4}', (0, Fr(5, 8), Fr(-5, 8), Fr(5, 4), Fr(-5, 4)), 4),
             ('F(4x4) {0,1/2,-1/2,5/4,-5/4}', (0, Fr(1, 2), Fr(-1, 2), Fr(5, 4), Fr(-5, 4)), 4),
             ('F(3x3) {0,1,-1,2}', (0, 1, -1, 2), 3),
             ('F(3x3) {0,1,-1,1/2}', (0, 1, -1, Fr(1, 2)), 3),
             ]
    for name, pts, m in cases:
        res = [run(pts, m, C, N, seed=s) for s in range(3)]
        sc = np.mean([r[0] for r in res])
        print(f'{name:36s} scale {sc:5.2f}  wino max {max(r[1] for r in res):.2e} rms {np.mean([r[2] for r in res]):.2e}   '
              f'direct max {max(r[3] for r in res):.2e} rms {np.mean([r[4] for r in res]):.2e}')
