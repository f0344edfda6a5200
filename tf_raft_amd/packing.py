"""Host-side repacking of Keras-layout weights into the layouts the HIP kernels consume.

Packed conv layout (include/raft_hip.h): for a Keras kernel ``(kh, kw, Cin, Cout)`` whose input
is the concatenation of channel groups ``sources = [(c_real, c_pad), ...]``::

    wp[t, k // 4, n, k % 4] = kernel[t // kw, t % kw, k_real, n]

``k`` runs over the padded channel axis (each source padded with zero rows to ``c_pad``, a
multiple of 32) and ``n`` over ``npad`` (Cout rounded up to a multiple of 64, zero columns).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv(kernel: np.ndarray, bias: np.ndarray,
              sources: Sequence[Tuple[int, int]] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """Return (wp float32[T, Kpad/4, npad, 4], bias float32[npad], npad)."""
    kernel = np.asarray(kernel, dtype=np.float32)
    kh, kw, cin, cout = kernel.shape
    if sources is None:
        sources = [(cin, round_up(cin, 32))]
    if sum(c for c, _ in sources) != cin:
        raise ValueError(f'sources {sources} do not add up to Cin={cin}')
    for _, cp in sources:
        if cp % 32:
            raise ValueError('padded source channel counts must be multiples of 32')
    kpad = sum(cp for _, cp in sources)
    npad = round_up(cout, 64)
    full = np.zeros((kh * kw, kpad, npad), dtype=np.float32)
    k_src = 0
    k_dst = 0
    flat = kernel.reshape(kh * kw, cin, cout)
    for c, cp in sources:
        full[:, k_dst:k_dst + c, :cout] = flat[:, k_src:k_src + c, :]
        k_src += c
        k_dst += cp
    wp = full.reshape(kh * kw, kpad // 4, 4, npad).transpose(0, 1, 3, 2)
    b = np.zeros((npad,), dtype=np.float32)
    b[:cout] = np.asarray(bias, dtype=np.float32)
    return np.ascontiguousarray(wp), b, npad


def dgrad_kernel(kernel: np.ndarray) -> np.ndarray:
    """Kernel of the INPUT gradient of a stride-1 'same' Conv2D with odd kernel sizes: dx = conv2d(dy, K') with
    K'[ky, kx, co, ci] = K[kh-1-ky, kw-1-kx, ci, co] (spatial flip, in/out transposed) -- the backward of reference
    update.py:10-11, 91-95, 138-140 runs on the forward convolution kernels (``raft_conv2d_f32``)."""
    k = np.asarray(kernel, dtype=np.float32)
    if k.ndim != 4 or k.shape[0] % 2 == 0 or k.shape[1] % 2 == 0:
        raise ValueError(f'dgrad_kernel expects an odd-sized (kh, kw, Cin, Cout) kernel, got {k.shape}')
    return np.ascontiguousarray(k[::-1, ::-1].transpose(0, 1, 3, 2))


def pack_conv_dgrad(kernel: np.ndarray, sources: Sequence[Tuple[int, int]] = None):
    """``pack_conv`` of ``dgrad_kernel(kernel)`` with a zero bias: (wp, bias, npad) for ``raft_conv2d_f32`` applied to dy."""
    dk = dgrad_kernel(kernel)
    return pack_conv(dk, np.zeros((dk.shape[3],), np.float32), sources)


# Winograd F(2x2, 3x3) weight transform (Lavin & Gray 2016): U = G g G^T
_WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=np.float64)


def winograd_kernel(kernel: np.ndarray) -> np.ndarray:
    """(3, 3, Cin, Cout) -> (4, 4, Cin, Cout): U[ty, tx] = sum_{u,v} G[ty, u] G[tx, v] g[u, v], evaluated in float64
    and rounded once to float32 (the entries of G are dyadic, so only the additions round)."""
    k = np.asarray(kernel, dtype=np.float64)
    if k.shape[:2] != (3, 3):
        raise ValueError(f'winograd_kernel expects a 3x3 kernel, got {k.shape[:2]}')
    return np.einsum('au,bv,uvio->abio', _WINO_G, _WINO_G, k).astype(np.float32)


def pack_conv_winograd(kernel: np.ndarray, bias: np.ndarray,
                       sources: Sequence[Tuple[int, int]] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """``pack_conv`` of the winograd-transformed kernel: the packed layout with 16 taps, consumed by
    ``conv_wino_kernel`` (csrc/conv_wino.h)."""
    return pack_conv(winograd_kernel(kernel), bias, sources)


# Winograd F(4x4, 3x3), interpolation points {0, +-5/8, +-3/2, inf} (csrc/conv_wino4.h).  Point order = tap order
# [0, +a, -a, +b, -b, inf].  Cook-Toom construction: A^T[j][i] = p_i^j (+ the x^(m-1) row of infinity), G[i][k] = p_i^k / N_i
# with N_i = prod_{j != i}(p_i - p_j), B^T[i] = coefficients of prod_{j != i}(x - p_j) (infinity: of prod_j (x - p_j)).
# The points were chosen by measured fp32 error against the float64 convolution (tools/study/wino_error.py): rms 1.1e-6
# at K = 256 against 2.3e-6 for the textbook {0, +-1, +-2} and 3.8e-7 for F(2x2, 3x3).
WINO4_A, WINO4_B = 0.625, 1.5


def cook_toom(points, m: int, r: int):
    """(A^T (m, n), G (n, r), B^T (n, n)) in float64, exact rational arithmetic inside, for the finite `points`
    (n - 1 of them, n = m + r - 1) plus the point at infinity (last row / column)."""
    from fractions import Fraction as Fr
    n = m + r - 1
    if len(points) != n - 1:
        raise ValueError(f'F({m}, {r}) needs {n - 1} finite points, got {len(points)}')
    pts = [Fr(x) for x in points]

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    at = [[pts[i] ** j for i in range(n - 1)] + [Fr(int(j == m - 1))] for j in range(m)]
    g, bt = [], []
    for i in range(n - 1):
        norm, poly = Fr(1), [Fr(1)]
        for j in range(n - 1):
            if j != i:
                norm *= pts[i] - pts[j]
                poly = polymul(poly, [-pts[j], Fr(1)])
        g.append([pts[i] ** k / norm for k in range(r)])
        bt.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for j in range(n - 1):
        poly = polymul(poly, [-pts[j], Fr(1)])
    g.append([Fr(0)] * (r - 1) + [Fr(1)])
    bt.append(poly)
    as_f64 = lambda mat: np.array([[float(x) for x in row] for row in mat], dtype=np.float64)
    return as_f64(at), as_f64(g), as_f64(bt)


WINO4_AT, _WINO4_G, WINO4_BT = cook_toom((0, WINO4_A, -WINO4_A, WINO4_B, -WINO4_B), 4, 3)


def winograd4_kernel(kernel: np.ndarray) -> np.ndarray:
    """(3, 3, Cin, Cout) -> (6, 6, Cin, Cout): U[ty, tx] = sum_{u,v} G[ty, u] G[tx, v] g[u, v] in float64, rounded once."""
    k = np.asarray(kernel, dtype=np.float64)
    if k.shape[:2] != (3, 3):
        raise ValueError(f'winograd4_kernel expects a 3x3 kernel, got {k.shape[:2]}')
    return np.einsum('au,bv,uvio->abio', _WINO4_G, _WINO4_G, k).astype(np.float32)


def wino4_tap_of_slot(q: int):
    """Slot q (0..71) of a 16-channel chunk in conv_wino4_kernel -> (tap row, tap column, half): the loop walks two halves
    (k-steps 2h, 2h + 1), each three phases of two tap rows -- (+a, -a), (+b, -b), (0, inf) -- of six taps."""
    inst, u = divmod(q, 12)
    h, ph = divmod(inst, 3)
    row, tx = divmod(u, 6)
    ty = (1 + row) if ph == 0 else ((3 + row) if ph == 1 else 5 * row)
    return ty, tx, h


def pack_conv_winograd4(kernel: np.ndarray, bias: np.ndarray,
                        sources: Sequence[Tuple[int, int]] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """F(4x4, 3x3)-transformed kernel in the order ``conv_wino4_kernel`` (csrc/conv_wino4.h) consumes it:
    wp float32[Kpad/16, 72, 4, npad/32, 16, 2, 2] = [16-channel chunk][slot q][k-quad G][n / 32][n % 16][(n / 16) % 2][e]
    holding U[tap(q)][16 c + 4 G + 2 h(q) + e][n] -- a lane's fragments of a slot for both column blocks of its wave are one
    16-byte load and a wave's weight stream is one constant stride per slot.
    ``sources`` as in ``pack_conv`` (padded channel counts: multiples of 16).  Returns (wp, bias[npad], npad)."""
    u = winograd4_kernel(kernel)
    _, _, cin, cout = u.shape
    if sources is None:
        sources = [(cin, round_up(cin, 16))]
    if sum(c for c, _ in sources) != cin:
        raise ValueError(f'sources {sources} do not add up to Cin={cin}')
    for _, cp in sources:
        if cp % 16:
            raise ValueError('padded source channel counts must be multiples of 16')
    kpad = sum(cp for _, cp in sources)
    npad = round_up(cout, 64)
    full = np.zeros((6, 6, kpad, npad), dtype=np.float32)
    k_src = k_dst = 0
    for c, cp in sources:
        full[:, :, k_dst:k_dst + c, :cout] = u[:, :, k_src:k_src + c, :]
        k_src += c
        k_dst += cp
    nch = kpad // 16
    wp = np.zeros((nch, 72, 4, npad // 32, 16, 2, 2), dtype=np.float32)
    for q in range(72):
        ty, tx, h = wino4_tap_of_slot(q)
        blk = full[ty, tx].reshape(nch, 4, 4, npad // 32, 2, 16)          # [chunk][G][channel in quad][n / 32][(n / 16) % 2][n % 16]
        wp[:, q] = blk[:, :, 2 * h:2 * h + 2].transpose(0, 1, 3, 5, 4, 2)  # [chunk][G][n / 32][n % 16][(n / 16) % 2][e]
    b = np.zeros((npad,), dtype=np.float32)
    b[:cout] = np.asarray(bias, dtype=np.float32)
    return wp, b, npad


def wino4_transform_6(d, a: float = WINO4_A, b: float = WINO4_B):
    """The 6-point transform B^T d along axis 0 exactly as conv_wino4.h evaluates it (pairs +-a, +-b share their even /
    odd parts): returns rows in tap order [0, +a, -a, +b, -b, inf].  Works on any array type with + - * (tests feed
    float32 to predict the kernel's rounding)."""
    a2, b2 = a * a, b * b
    s, p = a2 + b2, a2 * b2
    r0 = (d[4] - s * d[2]) + p * d[0]
    ta1, ta2 = d[4] - b2 * d[2], d[3] - b2 * d[1]
    tb1, tb2 = d[4] - a2 * d[2], d[3] - a2 * d[1]
    r5 = (d[5] - s * d[3]) + p * d[1]
    return [r0, ta1 + a * ta2, ta1 - a * ta2, tb1 + b * tb2, tb1 - b * tb2, r5]


def wino4_output_4(mm, a: float = WINO4_A, b: float = WINO4_B):
    """A^T m along axis 0 (6 taps in tap order -> 4 outputs) as conv_wino4.h evaluates it."""
    sa, da, sb, db = mm[1] + mm[2], mm[1] - mm[2], mm[3] + mm[4], mm[3] - mm[4]
    return [mm[0] + (sa + sb), a * da + b * db, (a * a) * sa + (b * b) * sb, ((a * a * a) * da + (b * b * b) * db) + mm[5]]


# 1-D Winograd F(2, 5), points {0, +-1, +-1/2, inf}, rows rescaled by powers of two (csrc/conv_wino1d.h): U = G' g
_WINO1D_G = np.array([[1.0, 0.0, 0.0, 0.0, 0.0],
                      [1 / 6, 1 / 6, 1 / 6, 1 / 6, 1 / 6],
                      [1 / 6, -1 / 6, 1 / 6, -1 / 6, 1 / 6],
                      [-4 / 3, -2 / 3, -1 / 3, -1 / 6, -1 / 12],
                      [-4 / 3, 2 / 3, -1 / 3, 1 / 6, -1 / 12],
                      [0.0, 0.0, 0.0, 0.0, 0.25]], dtype=np.float64)
WINO1D_BT = np.array([[1, 0, -5, 0, 4, 0], [0, -1, -1, 4, 4, 0], [0, 1, -1, -4, 4, 0],
                      [0, -1, -2, 1, 2, 0], [0, 1, -2, -1, 2, 0], [0, 1, 0, -5, 0, 4]], dtype=np.float64)
WINO1D_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 0.5, -0.5, 1]], dtype=np.float64)


# 1-D Winograd F(4, 5), points {0, +-1, +-1/2, +-2, inf}; B^T rows rescaled by 4, 4, 4, 2, 2, 4, 4, 4 to integers
# and G rows by the inverse (csrc/conv_wino1d.h, MO = 4)
_WINO1D4_G = np.array([[-1, 0, 0, 0, 0],
                       [-2 / 9, -2 / 9, -2 / 9, -2 / 9, -2 / 9],
                       [-2 / 9, 2 / 9, -2 / 9, 2 / 9, -2 / 9],
                       [32 / 45, 16 / 45, 8 / 45, 4 / 45, 2 / 45],
                       [32 / 45, -16 / 45, 8 / 45, -4 / 45, 2 / 45],
                       [1 / 90, 1 / 45, 2 / 45, 4 / 45, 8 / 45],
                       [1 / 90, -1 / 45, 2 / 45, -4 / 45, 8 / 45],
                       [0, 0, 0, 0, 1]], dtype=np.float64) / np.array([4, 4, 4, 2, 2, 4, 4, 4], dtype=np.float64)[:, None]
WINO1D4_BT = np.array([[-4, 0, 21, 0, -21, 0, 4, 0], [0, 4, 4, -17, -17, 4, 4, 0], [0, -4, 4, 17, -17, -4, 4, 0],
                       [0, 4, 8, -5, -10, 1, 2, 0], [0, -4, 8, 5, -10, -1, 2, 0], [0, 2, 1, -10, -5, 8, 4, 0],
                       [0, -2, 1, 10, -5, -8, 4, 0], [0, -4, 0, 21, 0, -21, 0, 4]], dtype=np.float64)
WINO1D4_AT = np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 0.5, -0.5, 2, -2, 0],
                       [0, 1, 1, 0.25, 0.25, 4, 4, 0], [0, 1, -1, 0.125, -0.125, 8, -8, 1]], dtype=np.float64)


def winograd1d_kernel(kernel: np.ndarray, m: int = 2) -> np.ndarray:
    """(1, 5, Cin, Cout) or (5, 1, Cin, Cout) -> (m + 4, 1, Cin, Cout): U[t] = sum_k G'[t, k] g[k], float64 then one
    rounding.  m = 2: F(2, 5), m = 4: F(4, 5)."""
    if m not in (2, 4):
        raise ValueError(f'winograd1d_kernel: m must be 2 or 4, got {m}')
    k = np.asarray(kernel, dtype=np.float64)
    if k.shape[:2] == (1, 5):
        g = k[0]
    elif k.shape[:2] == (5, 1):
        g = k[:, 0]
    else:
        raise ValueError(f'winograd1d_kernel expects a 1x5 or 5x1 kernel, got {k.shape[:2]}')
    return np.einsum('tk,kio->tio', _WINO1D_G if m == 2 else _WINO1D4_G, g).astype(np.float32)[:, None]


def pack_conv_winograd1d(kernel: np.ndarray, bias: np.ndarray, sources: Sequence[Tuple[int, int]] = None,
                         m: int = 2) -> Tuple[np.ndarray, np.ndarray, int]:
    """``pack_conv`` of the F(m, 5)-transformed kernel (m + 4 taps), consumed by ``conv_wino1d_kernel``."""
    return pack_conv(winograd1d_kernel(kernel, m), bias, sources)


def fuse_n(weights: Dict[str, np.ndarray], names: Sequence[str]):
    """Concatenate several convolutions over the same input along the output-channel axis."""
    k = np.concatenate([weights[f'{n}/kernel'] for n in names], axis=3)
    b = np.concatenate([weights[f'{n}/bias'] for n in names], axis=0)
    return k, b


def pack_convc1_fused(kernel: np.ndarray, bias: np.ndarray, levels: int = 4, radius: int = 4):
    """convc1 (1, 1, levels*(2r+1)^2, 256) for ``raft_lookup_convc1_f32``: K reordered level-major with each level's
    (2r+1)^2 = 81 rows padded to 84 (zero rows), packed as k-quads: (levels * 21, 256, 4)."""
    k = np.asarray(kernel, dtype=np.float32)
    d2 = (2 * radius + 1) ** 2
    if k.shape[:3] != (1, 1, levels * d2) or k.shape[3] != 256 or radius != 4 or levels != 4:
        raise ValueError(f'pack_convc1_fused expects a (1, 1, 324, 256) kernel, got {k.shape}')
    lvl = 84
    full = np.zeros((levels * lvl, 256), dtype=np.float32)
    for l in range(levels):
        full[l * lvl:l * lvl + d2] = k[0, 0, l * d2:(l + 1) * d2]
    wp = full.reshape(levels * lvl // 4, 4, 256).transpose(0, 2, 1)
    return np.ascontiguousarray(wp), np.asarray(bias, dtype=np.float32).copy(), 256


def pack_basic_update(weights: Dict[str, np.ndarray], prefix: str = 'update_block') -> List[Tuple[str, np.ndarray, np.ndarray, int]]:
    """[(field, packed_kernel, bias, npad)] for ``raft_basic_update_weights``
    (reference update.py:128-153; GRU input order hx = [h | inp | motion(126) | flow(2)])."""
    p = prefix
    w = weights
    out = []

    def conv(field, name, sources=None):
        wp, b, npad = pack_conv(w[f'{name}/kernel'], w[f'{name}/bias'], sources)
        out.append((field, wp, b, npad))

    conv('convc1', f'{p}/encoder/convc1', [(324, 352)])
    conv('convc2', f'{p}/encoder/convc2')
    out.append(('convf1', np.ascontiguousarray(w[f'{p}/encoder/convf1/kernel'], dtype=np.float32).reshape(98, 128),
                np.asarray(w[f'{p}/encoder/convf1/bias'], dtype=np.float32), 128))
    conv('convf2', f'{p}/encoder/convf2')
    conv('conv', f'{p}/encoder/conv')
    # SepConvGRU: the `inp` rows (128:256 of hx) of z / r / q go to the loop-invariant context
    # convolution gru_ctx{s} together with the biases; the per-iteration kernels keep [h | motion | flow]
    loop_rows = np.r_[0:128, 256:384]
    ctx, ctx_w4, gru_w, gru_w4 = [], [], [], []
    for s in ('1', '2'):
        k, b = fuse_n(w, [f'{p}/gru/convz{s}', f'{p}/gru/convr{s}'])
        wp, bb, npad = pack_conv(k[:, :, loop_rows, :], np.zeros_like(b), [(128, 128), (128, 128)])
        out.append((f'gru_zr{s}', wp, bb, npad))
        wp, bb, npad = pack_conv_winograd1d(k[:, :, loop_rows, :], np.zeros_like(b), [(128, 128), (128, 128)])
        gru_w.append((f'gru_zr{s}_w', wp, bb, npad))
        wp, bb, npad = pack_conv_winograd1d(k[:, :, loop_rows, :], np.zeros_like(b), [(128, 128), (128, 128)], m=4)
        gru_w4.append((f'gru_zr{s}_w4', wp, bb, npad))
        kq, bq = w[f'{p}/gru/convq{s}/kernel'], w[f'{p}/gru/convq{s}/bias']
        wp, bb, npad = pack_conv(kq[:, :, loop_rows, :], np.zeros_like(bq), [(128, 128), (128, 128)])
        out.append((f'gru_q{s}', wp, bb, npad))
        wp, bb, npad = pack_conv_winograd1d(kq[:, :, loop_rows, :], np.zeros_like(bq), [(128, 128), (128, 128)])
        gru_w.append((f'gru_q{s}_w', wp, bb, npad))
        wp, bb, npad = pack_conv_winograd1d(kq[:, :, loop_rows, :], np.zeros_like(bq), [(128, 128), (128, 128)], m=4)
        gru_w4.append((f'gru_q{s}_w4', wp, bb, npad))
        kc = np.concatenate([k, kq], axis=3)[:, :, 128:256, :]
        wp, bb, npad = pack_conv(kc, np.concatenate([b, bq]))
        ctx.append((f'gru_ctx{s}', wp, bb, npad))
        wp, bb, npad = pack_conv_winograd1d(kc, np.concatenate([b, bq]), m=4)
        ctx_w4.append((f'gru_ctx{s}_w4', wp, bb, npad))
    k, b = fuse_n(w, [f'{p}/flow_head/conv1', f'{p}/mask/0'])
    wp, bb, npad = pack_conv(k, b)
    out.append(('fh1_mask0', wp, bb, npad))
    out.append(('fh2', np.ascontiguousarray(w[f'{p}/flow_head/conv2/kernel'], dtype=np.float32).reshape(9, 256, 2),
                np.asarray(w[f'{p}/flow_head/conv2/bias'], dtype=np.float32), 2))
    conv('mask2', f'{p}/mask/2')
    out = out + ctx
    # winograd-transformed copies of the 3x3 layers (same order of fields as raft_basic_update_weights)
    k, b = fuse_n(w, [f'{p}/flow_head/conv1', f'{p}/mask/0'])
    for field, kk, bb_, src in (('convc2_w', w[f'{p}/encoder/convc2/kernel'], w[f'{p}/encoder/convc2/bias'], None),
                                ('convf2_w', w[f'{p}/encoder/convf2/kernel'], w[f'{p}/encoder/convf2/bias'], None),
                                ('conv_w', w[f'{p}/encoder/conv/kernel'], w[f'{p}/encoder/conv/bias'], None),
                                ('fh1_mask0_w', k, b, None)):
        wp, bb, npad = pack_conv_winograd(kk, bb_, src)
        out.append((field, wp, bb, npad))
    # F(2, 5)-transformed copies of the per-iteration SepConvGRU convolutions (field order of the C struct)
    order = ['gru_zr1_w', 'gru_q1_w', 'gru_zr2_w', 'gru_q2_w']
    out = out + sorted(gru_w, key=lambda e: order.index(e[0]))
    # flow_head.conv1 alone (final-only prediction loop: the mask half of fh1_mask0 is skipped)
    wp, bb, npad = pack_conv_winograd(w[f'{p}/flow_head/conv1/kernel'], w[f'{p}/flow_head/conv1/bias'])
    out.append(('fh1_w', wp, bb, npad))
    # F(4, 5)-transformed copies of the SepConvGRU convolutions
    order4 = ['gru_zr1_w4', 'gru_q1_w4', 'gru_zr2_w4', 'gru_q2_w4']
    out = out + sorted(gru_w4, key=lambda e: order4.index(e[0]))
    wp, bb, npad = pack_convc1_fused(w[f'{p}/encoder/convc1/kernel'], w[f'{p}/encoder/convc1/bias'])
    out.append(('convc1_f', wp, bb, npad))
    out = out + ctx_w4
    # Winograd F(4x4, 3x3) copies (csrc/conv_wino4.h); field order of the C struct
    k, b = fuse_n(w, [f'{p}/flow_head/conv1', f'{p}/mask/0'])
    for field, kk, bb_ in (('convc2_w44', w[f'{p}/encoder/convc2/kernel'], w[f'{p}/encoder/convc2/bias']),
                           ('conv_w44', w[f'{p}/encoder/conv/kernel'], w[f'{p}/encoder/conv/bias']),
                           ('fh1_mask0_w44', k, b),
                           ('fh1_w44', w[f'{p}/flow_head/conv1/kernel'], w[f'{p}/flow_head/conv1/bias']),
                           ('convf2_w44', w[f'{p}/encoder/convf2/kernel'], w[f'{p}/encoder/convf2/bias'])):
        wp, bb, npad = pack_conv_winograd4(kk, bb_)
        out.append((field, wp, bb, npad))
    return out


def pack_small_update(weights: Dict[str, np.ndarray], prefix: str = 'update_block'):
    """[(field, packed_kernel, bias, npad)] for ``raft_small_update_weights``
    (reference update.py:109-125; hx = [h(96) | inp(64) | motion(80) | flow(2)], x padded to 160)."""
    p = prefix
    w = weights
    out = []

    def conv(field, name, sources=None):
        wp, b, npad = pack_conv(w[f'{name}/kernel'], w[f'{name}/bias'], sources)
        out.append((field, wp, b, npad))

    conv('convc1', f'{p}/encoder/convc1', [(196, 224)])
    out.append(('convf1', np.ascontiguousarray(w[f'{p}/encoder/convf1/kernel'], dtype=np.float32).reshape(98, 64),
                np.asarray(w[f'{p}/encoder/convf1/bias'], dtype=np.float32), 64))
    conv('convf2', f'{p}/encoder/convf2')
    conv('conv', f'{p}/encoder/conv')
    k, b = fuse_n(w, [f'{p}/gru/convz', f'{p}/gru/convr'])
    wp, bb, npad = pack_conv(k, b, [(96, 96), (146, 160)])
    out.append(('gru_zr', wp, bb, npad))
    wp, bb, npad = pack_conv(w[f'{p}/gru/convq/kernel'], w[f'{p}/gru/convq/bias'], [(96, 96), (146, 160)])
    out.append(('gru_q', wp, bb, npad))
    conv('fh1', f'{p}/flow_head/conv1')
    out.append(('fh2', np.ascontiguousarray(w[f'{p}/flow_head/conv2/kernel'], dtype=np.float32).reshape(9, 128, 2),
                np.asarray(w[f'{p}/flow_head/conv2/bias'], dtype=np.float32), 2))
    # Winograd F(2x2, 3x3) copies of the 3x3 layers (field order of raft_small_update_weights)
    for field, kk, bb_, src in (('conv_w', w[f'{p}/encoder/conv/kernel'], w[f'{p}/encoder/conv/bias'], None),
                                ('gru_zr_w', k, b, [(96, 96), (146, 160)]),
                                ('gru_q_w', w[f'{p}/gru/convq/kernel'], w[f'{p}/gru/convq/bias'], [(96, 96), (146, 160)]),
                                ('fh1_w', w[f'{p}/flow_head/conv1/kernel'], w[f'{p}/flow_head/conv1/bias'], None)):
        wp, bb, npad = pack_conv_winograd(kk, bb_, src)
        out.append((field, wp, bb, npad))
    return out


# ------------------------------------------------------------------------------------------------
# encoders (reference extractor.py:88-175)
# ------------------------------------------------------------------------------------------------
BN_EPS = 1e-3   # Keras BatchNormalization default used by the reference (extractor.py:10)


def pack_stem(kernel: np.ndarray, bias: np.ndarray):
    """7x7 stride-2 stem over the 4-channel padded image: K chunk c = kernel row c, k = kx * 4 + ch
    (kx < 7, ch < 3; the rest zero).  Returns (wp float32[1, 56, npad, 4], bias[npad], npad)."""
    kernel = np.asarray(kernel, dtype=np.float32)
    kh, kw, cin, cout = kernel.shape
    if (kh, kw, cin) != (7, 7, 3):
        raise ValueError(f'stem kernel must be (7, 7, 3, Cout), got {kernel.shape}')
    npad = round_up(cout, 64)
    full = np.zeros((7, 8, 4, npad), dtype=np.float32)          # [ky][kx][ch][n]
    full[:, :7, :3, :cout] = kernel
    wp = full.reshape(1, 56, 4, npad).transpose(0, 1, 3, 2)      # [1][kq = ky*8 + kx][n][ch]
    b = np.zeros((npad,), dtype=np.float32)
    b[:cout] = np.asarray(bias, dtype=np.float32)
    return np.ascontiguousarray(wp), b, npad


def _fold_bn(kernel, bias, w, name):
    """Keras BatchNormalization (inference) folded into the preceding convolution:
    y = (conv + b - mean) * gamma / sqrt(var + eps) + beta."""
    if f'{name}/moving_mean' not in w:
        return kernel, bias
    s = (w[f'{name}/gamma'].astype(np.float64) / np.sqrt(w[f'{name}/moving_variance'].astype(np.float64) + BN_EPS))
    k = (kernel.astype(np.float64) * s).astype(np.float32)
    b = ((bias.astype(np.float64) - w[f'{name}/moving_mean'].astype(np.float64)) * s
         + w[f'{name}/beta'].astype(np.float64)).astype(np.float32)
    return k, b


def pack_encoder(weights: Dict[str, np.ndarray], prefix: str, norm_type):
    """Returns (convs, norms, dims): convs = [(field, wp, bias, npad)] with field in
    {'conv1', 'conv2', ('block', i, j), ('block_w', i, j), ('block_w44', i, j)} (block_w / block_w44: Winograd F(2x2) /
    F(4x4) transformed copies of a stride-1 3x3 convolution); norms = [(index, gamma, beta)] for instance norm;
    dims = (c0, c1, c2, c3, cout).  Batch norm is folded into the convolutions."""
    w = {k[len(prefix) + 1:]: np.asarray(v, dtype=np.float32) for k, v in weights.items() if k.startswith(prefix + '/')}
    convs, norms = [], []

    def conv(field, name, norm_name, stem=False, wino_field=None, wino4_field=None):
        k, b = w[f'{name}/kernel'], w[f'{name}/bias']
        if norm_type == 'batch' and norm_name is not None:
            k, b = _fold_bn(k, b, w, norm_name)
        wp, bb, npad = pack_stem(k, b) if stem else pack_conv(k, b)
        convs.append((field, wp, bb, npad))
        if wino_field is not None and k.shape[2] % 16 == 0:      # stride-1 3x3: also the Winograd F(2x2, 3x3) form
            wpw, bw, npw = pack_conv_winograd(k, b)
            convs.append((wino_field, wpw, bw, npw))
        if wino4_field is not None and k.shape[2] % 16 == 0:     # and the F(4x4, 3x3) form (csrc/conv_wino4.h)
            wp4, b4, np4 = pack_conv_winograd4(k, b)
            convs.append((wino4_field, wp4, b4, np4))

    def inorm(index, name):
        if norm_type == 'instance':
            norms.append((index, w[f'{name}/gamma'], w[f'{name}/beta']))

    conv('conv1', 'conv1', 'norm1', stem=True)
    inorm(0, 'norm1')
    dims = [w['conv1/kernel'].shape[3]]
    for li in (1, 2, 3):
        for bi in (0, 1):
            blk = (li - 1) * 2 + bi
            name = f'layer{li}/{bi}'
            strided = f'{name}/downsample/0/kernel' in w          # the block's first convolution has stride 2
            conv(('block', blk, 0), f'{name}/conv1', f'{name}/norm1', wino_field=None if strided else ('block_w', blk, 0),
                 wino4_field=None if strided else ('block_w44', blk, 0))
            conv(('block', blk, 1), f'{name}/conv2', f'{name}/norm2', wino_field=('block_w', blk, 1), wino4_field=('block_w44', blk, 1))
            inorm(1 + blk * 3, f'{name}/norm1')
            inorm(2 + blk * 3, f'{name}/norm2')
            if f'{name}/downsample/0/kernel' in w:
                conv(('block', blk, 2), f'{name}/downsample/0', f'{name}/downsample/1')
                inorm(3 + blk * 3, f'{name}/downsample/1')
        dims.append(w[f'layer{li}/0/conv1/kernel'].shape[3])
    conv('conv2', 'conv2', None)
    dims.append(w['conv2/kernel'].shape[3])
    return convs, norms, tuple(dims)
