"""Host-side logic that needs no GPU: weight tables, packing, the C-ABI library (loads, exports
every symbol declared in include/raft_hip.h, validates arguments before touching a device),
loud failure without a GPU.  CPU only -- no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from tf_raft_amd import _ffi, packing
from tf_raft_amd import weights as wm


# ---------------------------------------------------------------- weights
def test_parameter_counts_match_the_reference_architecture():
    """SURVEY 8a parameter inventory (Keras layer shapes of reference model.py / update.py / extractor.py)."""
    w = wm.init_weights('raft', 0)
    assert wm.count_params(w) == 5263296
    assert wm.count_params(w, 'fnet') == 1069728
    assert wm.count_params(w, 'cnet') == 1072608
    assert wm.count_params(w, 'update_block') == 3120960
    s = wm.init_weights('small', 0)
    assert wm.count_params(s) == 1874130
    assert wm.count_params(s, 'update_block') == 876530


def test_default_weights_are_keras_defaults_and_seeded():
    a, b = wm.init_weights('raft', 7), wm.init_weights('raft', 7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    k = a['update_block/gru/convz1/kernel']
    assert k.shape == (1, 5, 384, 128)
    limit = np.sqrt(6.0 / (5 * 384 + 5 * 128))
    assert np.abs(k).max() <= limit and np.abs(k).max() > 0.9 * limit
    assert not a['update_block/gru/convz1/bias'].any()
    assert np.all(a['cnet/norm1/moving_variance'] == 1) and np.all(a['fnet/norm1/gamma'] == 1)
    p = wm.init_weights('raft', 7, perturb=True)
    assert p['update_block/gru/convz1/bias'].any()
    wm.check_weights('raft', a)
    with pytest.raises(ValueError):
        wm.check_weights('small', a)
    bad = dict(a)
    bad['fnet/conv1/kernel'] = np.zeros((3, 3, 3, 64), np.float32)
    with pytest.raises(ValueError):
        wm.check_weights('raft', bad)


def test_save_load_roundtrip(tmp_path):
    w = wm.init_weights('small', 1, perturb=True)
    path = str(tmp_path / 'w.npz')
    wm.save_weights(path, w)
    r = wm.load_weights(path)
    assert list(r) == list(w) and all(np.array_equal(r[k], w[k]) for k in w)


def test_invalid_norm_type_raises_like_the_reference():
    with pytest.raises(ValueError, match='Invalid norm_type'):
        wm.encoder_entries('fnet', 'basic', 'layer', 128)         # reference extractor.py:16


# ---------------------------------------------------------------- packing
def _unpacked_conv(x, wp, bias, kh, kw, srcs_pad, nvalid):
    """Evaluate the packed layout exactly the way the kernel indexes it (NumPy, tiny sizes)."""
    B, H, W, kpad = x.shape
    T, kq, npad, four = wp.shape
    assert four == 4 and kq * 4 == kpad and T == kh * kw
    out = np.zeros((B, H, W, nvalid), np.float64)
    for t in range(T):
        dy, dx = t // kw - (kh - 1) // 2, t % kw - (kw - 1) // 2
        for k in range(kpad):
            wrow = wp[t, k // 4, :nvalid, k % 4]
            shifted = np.zeros((B, H, W))
            ys = slice(max(0, -dy), min(H, H - dy))
            xs = slice(max(0, -dx), min(W, W - dx))
            shifted[:, ys, xs] = x[:, ys.start + dy:ys.stop + dy, xs.start + dx:xs.stop + dx, k]
            out += shifted[..., None] * wrow
    return out + bias[:nvalid]


@pytest.mark.parametrize('ksize', [(1, 1), (3, 3), (1, 5), (5, 1)])
def test_pack_conv_layout_reproduces_the_convolution(rng, ksize):
    from oracle import tf_ops
    kh, kw = ksize
    c_a, c_b, cout = 5, 7, 6
    kernel = rng.normal(size=(kh, kw, c_a + c_b, cout)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    wp, b, npad = packing.pack_conv(kernel, bias, [(c_a, 32), (c_b, 32)])
    assert wp.shape == (kh * kw, 16, 64, 4) and npad == 64 and b.shape == (64,)
    xa = rng.normal(size=(1, 5, 6, c_a)).astype(np.float32)
    xb = rng.normal(size=(1, 5, 6, c_b)).astype(np.float32)
    xpad = np.zeros((1, 5, 6, 64), np.float32)
    xpad[..., :c_a] = xa
    xpad[..., 32:32 + c_b] = xb
    got = _unpacked_conv(xpad, wp, b, kh, kw, None, cout)
    want = tf_ops.conv2d(torch.as_tensor(np.concatenate([xa, xb], -1)).double(),
                         torch.as_tensor(kernel).double(), torch.as_tensor(bias).double()).numpy()
    np.testing.assert_allclose(got, want, atol=1e-5)
    assert not wp[:, :, cout:, :].any()                              # zero N padding


def test_pack_conv_rejects_bad_sources():
    k = np.zeros((1, 1, 8, 4), np.float32)
    with pytest.raises(ValueError):
        packing.pack_conv(k, np.zeros(4), [(5, 32)])
    with pytest.raises(ValueError):
        packing.pack_conv(k, np.zeros(4), [(8, 20)])


def test_pack_update_blocks_cover_every_struct_field():
    basic = packing.pack_basic_update({k: v for k, v in wm.init_weights('raft', 0).items()})
    assert [f for f, *_ in basic] == [n for n, _ in _ffi.BasicUpdateWeights._fields_]
    zr = dict((f, (wp, b, n)) for f, wp, b, n in basic)['gru_zr1']
    assert zr[0].shape == (5, 64, 256, 4) and zr[2] == 256           # [z | r] fused along N; K = h + [motion | flow]
    assert not zr[1].any()                                           # the biases ride in the context convolution
    ctx = dict((f, (wp, b, n)) for f, wp, b, n in basic)['gru_ctx1']
    assert ctx[0].shape == (5, 32, 384, 4) and ctx[2] == 384         # inp rows -> [z | r | q]
    # F(4x4, 3x3) copies in the kernel's consumption order: (Cin / 16, 72 slots, 4, npad / 32, 16, 2, 2)
    by = dict((f, (wp, b, n)) for f, wp, b, n in basic)
    for field, cin, cout in (('convc2_w44', 256, 192), ('convf2_w44', 128, 64), ('conv_w44', 256, 126), ('fh1_mask0_w44', 128, 512),
                             ('fh1_w44', 128, 256)):
        wp, b, npad = by[field]
        assert npad == packing.round_up(cout, 64) and wp.shape == (cin // 16, 72, 4, npad // 32, 16, 2, 2), field
        assert b.shape == (npad,) and not b[cout:].any()
    small = packing.pack_small_update(wm.init_weights('small', 0))
    assert [f for f, *_ in small] == [n for n, _ in _ffi.SmallUpdateWeights._fields_]
    szr = dict((f, (wp, b, n)) for f, wp, b, n in small)['gru_zr']
    assert szr[0].shape == (9, 64, 192, 4)                           # K = 96 + 160 (146 padded)


@pytest.mark.parametrize('s,ksize', [('1', (1, 5)), ('2', (5, 1))])
def test_gru_context_split_reproduces_the_full_convolution(rng, s, ksize):
    """convz/convr/convq over hx = [h | inp | motion | flow] (reference update.py:53-65) ==
    context convolution over inp (with the biases) + loop convolution over [h | motion | flow] (zero bias),
    evaluated from the packed arrays exactly as the kernels index them."""
    from oracle import tf_ops
    wts = wm.init_weights('raft', 3)
    for k in list(wts):
        if k.endswith('/bias'):
            wts[k] = rng.normal(size=wts[k].shape).astype(np.float32)
    packed = dict((f, (wp, b, n)) for f, wp, b, n in packing.pack_basic_update(wts))
    kh, kw = ksize
    hx = rng.normal(size=(1, 6, 7, 384)).astype(np.float32)
    p = 'update_block/gru'
    for field, names, nout in ((f'gru_zr{s}', [f'convz{s}', f'convr{s}'], 256), (f'gru_q{s}', [f'convq{s}'], 128)):
        kern = np.concatenate([wts[f'{p}/{n}/kernel'] for n in names], 3)
        bias = np.concatenate([wts[f'{p}/{n}/bias'] for n in names])
        want = tf_ops.conv2d(torch.as_tensor(hx).double(), torch.as_tensor(kern).double(),
                             torch.as_tensor(bias).double()).numpy()
        wp, b, npad = packed[field]
        loop_in = np.concatenate([hx[..., :128], hx[..., 256:]], -1)
        loop = _unpacked_conv(loop_in, wp, b, kh, kw, None, nout)
        cwp, cb, cn = packed[f'gru_ctx{s}']
        ctx = _unpacked_conv(hx[..., 128:256], cwp, cb, kh, kw, None, 384)
        ctx = ctx[..., :256] if nout == 256 else ctx[..., 256:384]
        np.testing.assert_allclose(loop + ctx, want, atol=1e-4)


def test_pack_stem_layout_reproduces_the_stride2_convolution(rng):
    """The 7x7/2 stem is packed as 7 K-chunks (one per kernel row) of k = kx*4 + ch over the 4-channel
    padded image; evaluate it exactly the way csrc/conv_halo.h (STEM mode) indexes it."""
    from oracle import tf_ops
    cout, H, W = 6, 12, 10
    kernel = rng.normal(size=(7, 7, 3, cout)).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    wp, b, npad = packing.pack_stem(kernel, bias)
    assert wp.shape == (1, 56, 64, 4) and npad == 64
    img = rng.normal(size=(1, H, W, 3)).astype(np.float32)
    img4 = np.zeros((1, H, W, 4), np.float64)
    img4[..., :3] = img
    (pt, _), (pl, _) = tf_ops.same_padding(H, 7, 2), tf_ops.same_padding(W, 7, 2)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    got = np.zeros((1, Ho, Wo, cout))
    for y in range(Ho):
        for x in range(Wo):
            for c in range(7):                       # K chunk = kernel row
                yy = 2 * y - pt + c
                for c4 in range(8):                  # 16-byte quad = kernel column (7 used)
                    xx = 2 * x - pl + c4
                    if c4 < 7 and 0 <= yy < H and 0 <= xx < W:
                        got[0, y, x] += img4[0, yy, xx] @ wp[0, c * 8 + c4, :cout, :].T.astype(np.float64)
    got += b[:cout]
    want = tf_ops.conv2d(torch.as_tensor(img).double(), torch.as_tensor(kernel).double(),
                         torch.as_tensor(bias).double(), stride=2).numpy()
    np.testing.assert_allclose(got, want, atol=1e-5)
    assert not wp[0, 7::8].any() and not wp[..., 3].any()             # zero padding column / channel


def test_pack_encoder_folds_batch_norm_and_covers_every_layer(rng):
    from oracle import tf_ops
    wts = wm.init_weights('raft', seed=1, perturb=True)
    convs, norms, dims = packing.pack_encoder(wts, 'cnet', 'batch')
    assert dims == (64, 64, 96, 128, 256) and norms == []
    fields = [f for f, *_ in convs]
    direct = [f for f in fields if not (isinstance(f, tuple) and f[0] in ('block_w', 'block_w44'))]
    assert direct[0] == 'conv1' and direct[-1] == 'conv2' and len(direct) == 1 + 6 * 2 + 2 + 1
    assert ('block', 2, 2) in fields and ('block', 4, 2) in fields and ('block', 0, 2) not in fields
    # Winograd copies: every stride-1 3x3 convolution (all conv2, and conv1 of the blocks without a down-sampling branch)
    wino = sorted(f for f in fields if isinstance(f, tuple) and f[0] == 'block_w')
    assert wino == sorted([('block_w', b, 1) for b in range(6)] + [('block_w', b, 0) for b in (0, 1, 3, 5)])
    wino4 = sorted(f for f in fields if isinstance(f, tuple) and f[0] == 'block_w44')
    assert wino4 == sorted([('block_w44', b, 1) for b in range(6)] + [('block_w44', b, 0) for b in (0, 1, 3, 5)])
    by_field = {f: (wp, bb, npad) for f, wp, bb, npad in convs}
    wp4, _, npad4 = by_field[('block_w44', 0, 1)]                      # F(4x4): (Cin/16, 72 slots, 4, npad/32, 16, 2, 2)
    assert wp4.shape == (4, 72, 4, 2, 16, 2, 2) and npad4 == 64
    wp_w, _, npad_w = by_field[('block_w', 3, 1)]
    assert wp_w.shape == (16, 96 // 4, npad_w, 4) and npad_w == 128
    # U = G g G^T: the four corner taps are the corner kernel entries, and summing U over taps reproduces sum(g) * 2.25
    name = 'cnet/layer2/1/conv2'
    kf3, _ = packing._fold_bn(wts[f'{name}/kernel'], wts[f'{name}/bias'],
                              {kk[len('cnet/'):]: v for kk, v in wts.items() if kk.startswith('cnet/')}, 'layer2/1/norm2')
    U = packing.winograd_kernel(kf3)
    np.testing.assert_array_equal(U[0, 0], kf3[0, 0])
    np.testing.assert_array_equal(U[3, 3], kf3[2, 2])
    np.testing.assert_allclose(U[1, 1], kf3.sum(axis=(0, 1)) / 4, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(wp_w[5].transpose(0, 2, 1).reshape(96, npad_w)[:, :96], U[1, 1], rtol=0, atol=0)
    # folded conv == conv followed by inference batch norm
    name = 'cnet/layer2/0/conv2'
    k, bias = wts[f'{name}/kernel'], wts[f'{name}/bias']
    n = 'cnet/layer2/0/norm2'
    kf, bf = packing._fold_bn(k, bias, {kk[len('cnet/'):]: v for kk, v in wts.items() if kk.startswith('cnet/')},
                              'layer2/0/norm2')
    x = torch.as_tensor(rng.normal(size=(1, 5, 6, k.shape[2]))).double()
    want = tf_ops.batch_norm(tf_ops.conv2d(x, torch.as_tensor(k).double(), torch.as_tensor(bias).double()),
                             *[torch.as_tensor(wts[f'{n}/{q}']).double() for q in ('gamma', 'beta', 'moving_mean',
                                                                                 'moving_variance')])
    got = tf_ops.conv2d(x, torch.as_tensor(kf).double(), torch.as_tensor(bf).double())
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-5)
    # instance norm: affine parameters are passed through, indexed stem = 0, block i -> 1 + 3 i + {0, 1, 2}
    convs, norms, dims = packing.pack_encoder(wts, 'fnet', 'instance')
    idx = sorted(i for i, _, _ in norms)
    assert idx == sorted([0] + [1 + 3 * b + j for b in range(6) for j in (0, 1)] + [1 + 3 * 2 + 2, 1 + 3 * 4 + 2])
    small = packing.pack_encoder(wm.init_weights('small', seed=1), 'cnet', None)
    assert small[2] == (32, 32, 64, 96, 160) and small[1] == []
    enc = _ffi.EncoderWeights()
    assert len(enc.in_gamma) == 19 and len(enc.block) == 6 and len(enc.block[0]) == 3
    assert len(enc.block_w) == 6 and len(enc.block_w[0]) == 2


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('ksize', [(1, 5), (5, 1)])
def test_winograd1d_transforms_reproduce_the_convolution(rng, m, ksize):
    """A^T [(G' g) . (B'^T d)] == the 5-tap correlation, for the F(2, 5) and F(4, 5) matrices the HIP kernel hard-codes
    (csrc/conv_wino1d.h) and the packed layout it reads."""
    kh, kw = ksize
    cin, cout = 32, 24
    kernel = rng.normal(size=(kh, kw, cin, cout))
    U = packing.winograd1d_kernel(kernel, m)
    assert U.shape == (m + 4, 1, cin, cout) and U.dtype == np.float32
    BT, AT = (packing.WINO1D_BT, packing.WINO1D_AT) if m == 2 else (packing.WINO1D4_BT, packing.WINO1D4_AT)
    assert BT.shape == (m + 4, m + 4) and AT.shape == (m, m + 4)
    assert np.all(BT == np.round(BT))                       # integer input transform: exact products in fp32
    d = rng.normal(size=(m + 4, cin))
    V = BT @ d                                              # (taps, cin)
    y = AT @ np.einsum('tc,tco->to', V, U[:, 0].astype(np.float64))
    g = kernel.reshape(5, cin, cout)
    want = np.stack([np.einsum('kc,kco->o', d[i:i + 5], g) for i in range(m)])
    np.testing.assert_allclose(y, want, atol=2e-5, rtol=0)  # U is rounded to fp32 once
    wp, b, npad = packing.pack_conv_winograd1d(kernel.astype(np.float32), np.zeros(cout, np.float32), [(cin, cin)], m=m)
    assert wp.shape == (m + 4, cin // 4, npad, 4) and npad % 32 == 0 and npad >= cout
    np.testing.assert_array_equal(wp[3].transpose(0, 2, 1).reshape(cin, npad)[:, :cout],
                                  packing.winograd1d_kernel(kernel.astype(np.float32), m)[3, 0])
    with pytest.raises(ValueError):
        packing.winograd1d_kernel(kernel, 3)
    with pytest.raises(ValueError):
        packing.winograd1d_kernel(rng.normal(size=(3, 3, 4, 4)), m)


# ---------------------------------------------------------------- C ABI
def _declared_functions():
    with open(os.path.join(ROOT, 'include', 'raft_hip.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(raft_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 20
    lib = C.CDLL(_ffi.library_path()) if os.path.exists(_ffi.library_path()) else None
    if lib is None:
        lib = C.CDLL(__import__('tf_raft_amd.build', fromlist=['x']).build_library(verbose=False))
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/raft_hip.h but not exported'
    assert sorted(_ffi.EXPORTED_SYMBOLS) == names, 'ctypes signature table and header disagree'
    typed = _ffi.load_library()
    assert typed.raft_version() == _ffi.ABI_VERSION
    with open(os.path.join(ROOT, 'include', 'raft_hip.h')) as f:
        assert int(re.search(r'#define RAFT_HIP_VERSION (\d+)', f.read()).group(1)) == _ffi.ABI_VERSION
    assert b'NULL' in typed.raft_error_string(-1)
    assert typed.raft_error_string(0) == b'ok'


def test_tuning_switches_are_a_table_not_getenv():
    """raft_set_option / raft_get_option (host-only): unknown names are rejected, values round-trip, NULL returns to
    the load-time state; and no launch path calls getenv (only the one-time table initialiser in host_util.hip may)."""
    assert _ffi.get_option('RAFT_GRU_WINO') == os.environ.get('RAFT_GRU_WINO', '')
    _ffi.set_option('RAFT_GRU_WINO', 5)
    assert _ffi.get_option('RAFT_GRU_WINO') == '5'
    _ffi.set_option('RAFT_GRU_WINO', '')
    assert _ffi.get_option('RAFT_GRU_WINO') == ''
    _ffi.set_option('RAFT_GRU_WINO', None)
    _ffi.set_option('RAFT_CONV_TILE', '256:5:171,128:5:141')
    assert _ffi.get_option('RAFT_CONV_TILE') == '256:5:171,128:5:141'
    _ffi.set_option('RAFT_CONV_TILE', None)
    with pytest.raises(ValueError):
        _ffi.set_option('RAFT_NO_SUCH_SWITCH', 1)
    csrc = os.path.join(ROOT, 'tf_raft_amd', 'csrc')
    for name in os.listdir(csrc):
        if name == 'host_util.hip':
            continue
        with open(os.path.join(csrc, name)) as f:
            assert 'getenv' not in re.sub(r'//.*', '', f.read()), f'{name} reads the environment on a launch path'


@pytest.mark.parametrize('path', ['tf_raft.model', 'tf_raft.losses', 'tf_raft.losses.losses', 'tf_raft.layers.corr',
                                  'tf_raft.layers.update', 'tf_raft.layers.extractor', 'tf_raft.training'])
def test_every_reference_import_path_resolves(path):
    """The reference's own import statements (train_sintel.py:8-9, tf_raft/model.py:4-6, tests/) work on the shim."""
    import importlib
    mod = importlib.import_module(path)
    names = {'tf_raft.model': ['RAFT', 'SmallRAFT'],
             'tf_raft.losses': ['sequence_loss', 'end_point_error', 'EndPointError'],
             'tf_raft.losses.losses': ['sequence_loss', 'end_point_error', 'EndPointError'],
             'tf_raft.layers.corr': ['CorrBlock', 'bilinear_sampler', 'coords_grid', 'upflow8'],
             'tf_raft.layers.update': ['BasicUpdateBlock', 'SmallUpdateBlock'],
             'tf_raft.layers.extractor': ['BasicEncoder', 'SmallEncoder'],
             'tf_raft.training': ['first_cycle_scaler', 'inverse_scaler']}[path]
    for n in names:
        assert hasattr(mod, n), (path, n)


def _header_struct_fields(name):
    """Field names of `typedef struct name { ... } name;` in include/raft_hip.h, in declaration order."""
    with open(os.path.join(ROOT, 'include', 'raft_hip.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    body = re.search(r'typedef struct %s\s*\{(.*?)\}\s*%s\s*;' % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith('const') else decl.split(None, 2)[2]
        for n in names.split(','):
            fields.append(re.sub(r'[\s\*]|\[.*?\]', '', n))
    return fields


@pytest.mark.parametrize('c_name,py_name', [('raft_conv_weights', 'ConvWeights'),
                                            ('raft_basic_update_weights', 'BasicUpdateWeights'),
                                            ('raft_small_update_weights', 'SmallUpdateWeights'),
                                            ('raft_encoder_weights', 'EncoderWeights'),
                                            ('raft_state', 'State')])
def test_ctypes_structs_mirror_the_header_field_for_field(c_name, py_name):
    """The ctypes mirrors in tf_raft_amd/_ffi.py declare the same fields in the same order as include/raft_hip.h
    (appending a field to one side only would silently shift every later pointer)."""
    want = _header_struct_fields(c_name)
    got = [f[0] for f in getattr(_ffi, py_name)._fields_]
    assert got == want


def test_pyramid_layout_host_helper():
    lib = _ffi.load_library()
    off = (C.c_int64 * 5)()
    lh, lw = (C.c_int * 4)(), (C.c_int * 4)()
    assert lib.raft_corr_pyramid_layout(1, 56, 64, 4, off, lh, lw) == 0
    assert list(lh) == [56, 28, 14, 7] and list(lw) == [64, 32, 16, 8]        # reference corr.py:112-114
    n = 56 * 64
    # per-query maps are stored as 4x8 tiles padded to whole tiles: 14x16 -> 4x2 tiles = 256, 7x8 -> 2x1 tiles = 64
    assert list(off) == [0, n * n, n * n + n * 896, n * n + n * (896 + 256), n * (n + 896 + 256 + 64)]
    assert lib.raft_corr_pyramid_layout(4, 8, 12, 4, off, lh, lw) == 0         # reference test size 64x96
    assert list(lh) == [8, 4, 2, 1] and list(lw) == [12, 6, 3, 1]
    nq = 4 * 8 * 12
    assert list(off) == [0, nq * 128, nq * (128 + 32), nq * (128 + 64), nq * (128 + 96)]
    assert lib.raft_corr_pyramid_layout(1, 4, 4, 4, off, lh, lw) == -2         # pooled away: RAFT_E_SHAPE
    assert lib.raft_corr_pyramid_layout(1, 8, 8, 5, off, lh, lw) == -3         # RAFT_E_UNSUPPORTED
    assert lib.raft_corr_build_workspace_floats(2, 56, 64, 256, 4) == 2 * 4800 * 256    # tile-padded rows
    assert lib.raft_update_workspace_floats(4, 56, 64) == 4 * 3584 * 1924      # + second [fh1 | mask.0] buffer, two flow copies


def test_tiled_map_layout_round_trip():
    """tile_maps / untile_maps (host mirror of common.h raft_tiled_index) on ragged sizes."""
    from tf_raft_amd.layers.corr import tile_maps, untile_maps
    g = torch.Generator().manual_seed(0)
    for h, w in [(56, 64), (14, 16), (7, 8), (3, 5), (1, 1), (5, 17)]:
        m = torch.rand((3, h, w), generator=g)
        t = tile_maps(m)
        ty, tx = (h + 3) // 4, (w + 7) // 8
        assert t.shape == (3, ty * tx * 32)
        assert torch.equal(untile_maps(t, h, w), m)
        y, x = h - 1, w - 1                                                     # the documented element formula
        assert t[1, ((y // 4) * tx + x // 8) * 32 + (y % 4) * 8 + x % 8] == m[1, y, x]
        assert float(t.sum()) == pytest.approx(float(m.sum()), rel=1e-6)        # padding is zero


def test_argument_errors_are_returned_before_any_device_work():
    lib = _ffi.load_library()
    off = (C.c_int64 * 5)(0, 1, 2, 3, 4)
    assert lib.raft_corr_lookup_f32(None, off, None, 1, 8, 8, 4, 4, None, 324, None) == -1
    assert lib.raft_coords_grid_f32(None, 1, 8, 8, None) == -1
    assert lib.raft_upsample_convex_f32(None, None, 1, 8, 8, None, None) == -1
    assert lib.raft_corr_build_f32(None, None, 1, 8, 8, 256, 4, None, off, None, None) == -1
    assert lib.raft_update_basic_f32(None, 1, 8, 8, None, None) == -1
    with pytest.raises(ValueError, match='NULL'):
        _ffi.check(-1, 'x')
    with pytest.raises(ValueError):
        _ffi.check(-2)


def test_product_path_fails_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import tf_raft_amd
    from tf_raft_amd.layers.corr import coords_grid
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        tf_raft_amd.RAFT()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        coords_grid(1, 4, 4)
    from tf_raft.losses.losses import end_point_error, sequence_loss   # the reference's import path
    gt, valid = np.zeros((1, 2, 2, 2), np.float32), np.ones((1, 2, 2), bool)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        sequence_loss((gt, valid), [gt])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        end_point_error((gt, valid), gt)
    from tf_raft_amd.prefetch import prefetch_to_device                 # the upload stage has nowhere to upload to
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        next(prefetch_to_device([(gt, gt)]))
    with pytest.raises(ValueError):
        next(prefetch_to_device([(gt, gt)], buffer_size=0))


def test_running_mean_mirrors_keras_mean():
    """tf.keras.metrics.Mean as the reference uses it (model.py:118-124, 168-170): mean of the fed scalars, 0 when empty."""
    from tf_raft_amd.losses import Mean
    m = Mean(name='epe')
    assert m.name == 'epe' and m.result() == 0.0
    for v in (1.0, 2.0, np.float32(6.0)):
        m.update_state(v)
    assert m.result() == 3.0 and m.count == 3
    m.reset_states()
    assert m.result() == 0.0 and m.count == 0


def test_deferred_weight_gradients_keep_one_geometry_per_layer():
    """grad.WgradDefer (the update block's weight gradients once per training step over all loop iterations): applications
    added under one key must share kernel shape and tensor geometry -- the multi-segment kernel indexes every segment alike --
    and a key keeps its trimming rule (padded channel counts are cut back when the gradients are assembled)."""
    import torch
    from tf_raft_amd import grad
    d = grad.WgradDefer()
    x, dy = torch.zeros((2, 6, 8, 16)), torch.zeros((2, 6, 8, 32))
    d.add('layer', x, dy, (3, 3, 16, 32))
    d.add('layer', x.clone(), dy.clone(), (3, 3, 16, 32))
    assert len(d.jobs['layer']['xs']) == 2 and d.jobs['layer']['trim'] is None
    with pytest.raises(ValueError, match='geometry'):
        d.add('layer', torch.zeros((2, 6, 9, 16)), dy, (3, 3, 16, 32))
    with pytest.raises(ValueError, match='geometry'):
        d.add('layer', x, dy, (1, 1, 16, 32))
    d.add(('zr', 'update_block', '1', 128), x, dy, (1, 5, 16, 32), trim=(14, 30))
    assert d.jobs[('zr', 'update_block', '1', 128)]['trim'] == (14, 30)
    assert grad.WgradDefer.MAX_SEG == 32                           # WG_MAXSEG of csrc/backward.hip: pointer pairs per launch


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under tf_raft_amd/ or tf_raft/ may reference it."""
    for pkg in ('tf_raft_amd', 'tf_raft'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for fn in files:
                if fn.endswith('.py'):
                    with open(os.path.join(dirpath, fn)) as f:
                        src = f.read()
                    assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, fn)


def test_a_stale_library_is_never_loaded_silently(monkeypatch):
    """if refreshing the build fails while an older libraft_hip.so exists, loading it must be an error (or an
    explicit opt-in) when it was built from other sources, and a warning when it is up to date."""
    from tf_raft_amd import _ffi, build

    def broken(**kw):
        raise RuntimeError('simulated compile error')
    monkeypatch.setattr(_ffi, '_LIB', None)
    monkeypatch.setattr(build, 'build_library', broken)
    monkeypatch.setattr(build, 'built_digest', lambda: 'digest-of-older-sources')
    monkeypatch.delenv('RAFT_ALLOW_STALE_LIB', raising=False)
    with pytest.raises(RuntimeError, match='built from different sources'):
        _ffi.load_library()
    monkeypatch.setenv('RAFT_ALLOW_STALE_LIB', '1')
    with pytest.warns(RuntimeWarning, match='STALE'):
        assert _ffi.load_library().raft_version() == _ffi.ABI_VERSION
    monkeypatch.setattr(_ffi, '_LIB', None)
    monkeypatch.delenv('RAFT_ALLOW_STALE_LIB')
    monkeypatch.setattr(build, 'built_digest', build.source_digest)
    with pytest.warns(RuntimeWarning, match='up-to-date'):
        _ffi.load_library()
    # a prebuilt .so that arrived WITHOUT its stamp cannot be verified: refused unless the caller opts in, then
    # loaded with a warning that says so; the ABI check still applies
    monkeypatch.setattr(_ffi, '_LIB', None)
    monkeypatch.setattr(build, 'built_digest', lambda: None)
    monkeypatch.delenv('RAFT_ALLOW_UNVERIFIED_LIB', raising=False)
    with pytest.raises(RuntimeError, match='no build stamp'):
        _ffi.load_library()
    monkeypatch.setenv('RAFT_ALLOW_UNVERIFIED_LIB', '1')
    with pytest.warns(RuntimeWarning, match='UNVERIFIED'):
        assert _ffi.load_library().raft_version() == _ffi.ABI_VERSION


def _wino4_emulate(x, wp, bias, cout):
    """What conv_wino4_kernel computes, restated in NumPy from the PACKED weight stream: per 16-channel chunk and slot q
    the tap (row, column, half) of packing.wino4_tap_of_slot, V = B^T d B by the kernel's two stages (fp32), products
    accumulated per tap in fp32, A^T . A at the end.  x: (H, W, Cpad) with H, W multiples of 4."""
    H, W, cpad = x.shape
    nch, npad = wp.shape[0], wp.shape[3] * 32
    wq = wp.transpose(0, 1, 2, 6, 3, 5, 4).reshape(nch, 72, 4, 2, npad)            # [chunk][slot][G][e][n]
    xp = np.zeros((H + 2, W + 2, cpad), np.float32)
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, cout), np.float32)
    for ty0 in range(0, H, 4):
        for tx0 in range(0, W, 4):
            d = xp[ty0:ty0 + 6, tx0:tx0 + 6]                                   # 6 x 6 x C patch
            w_rows = np.stack(packing.wino4_transform_6([d[r] for r in range(6)]))           # stage 1: over patch rows
            v = np.stack(packing.wino4_transform_6([w_rows[:, j] for j in range(6)]), axis=1)  # stage 2: [ty][tx][c]
            acc = np.zeros((6, 6, npad), np.float32)
            for c in range(nch):
                for q in range(72):
                    ty, tx, h = packing.wino4_tap_of_slot(q)
                    for g in range(4):
                        for e in range(2):
                            ch = 16 * c + 4 * g + 2 * h + e
                            acc[ty, tx] += v[ty, tx, ch] * wq[c, q, g, e]
            t = np.stack(packing.wino4_output_4([acc[i] for i in range(6)]))                 # over tap rows: [i][tx][n]
            y = np.stack(packing.wino4_output_4([t[:, j] for j in range(6)]), axis=1)        # [i][jx][n]
            out[ty0:ty0 + 4, tx0:tx0 + 4] = y[:, :, :cout] + bias[:cout]
    return out


def test_winograd4_transforms_and_packed_stream_reproduce_the_convolution(rng):
    """F(4x4, 3x3) with the points {0, +-5/8, +-3/2, inf}: the Cook-Toom matrices are exact, the pair-structured
    formulas the HIP kernel hard-codes (csrc/conv_wino4.h) equal B^T / A^T, and the packed weight stream walked in the
    kernel's slot order reproduces the 3x3 'same' convolution (two sources, padded channels)."""
    AT, G, BT = packing.cook_toom((0, packing.WINO4_A, -packing.WINO4_A, packing.WINO4_B, -packing.WINO4_B), 4, 3)
    assert AT.shape == (4, 6) and G.shape == (6, 3) and BT.shape == (6, 6)
    d = rng.normal(size=(6, 7))
    np.testing.assert_allclose(np.stack(packing.wino4_transform_6(list(d))), BT @ d, atol=1e-12)
    np.testing.assert_allclose(np.stack(packing.wino4_output_4(list(d))), AT @ d, atol=1e-12)
    # 1-D: y[i] = sum_k d[i + k] g[k]
    g = rng.normal(size=3)
    np.testing.assert_allclose(AT @ ((G @ g) * (BT @ d[:, 0])), [d[i:i + 3, 0] @ g for i in range(4)], atol=1e-12)
    # every B^T / A^T coefficient is exact in fp32 (dyadic points)
    assert np.all(BT.astype(np.float32) == BT) and np.all(AT.astype(np.float32) == AT)
    # slots: each (tap, half) exactly once
    slots = [packing.wino4_tap_of_slot(q) for q in range(72)]
    assert sorted(slots) == sorted((ty, tx, h) for ty in range(6) for tx in range(6) for h in range(2))
    # packed stream vs the convolution
    c_a, c_b, cout = 20, 16, 10
    kernel = (rng.normal(size=(3, 3, c_a + c_b, cout)) * 0.2).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    wp, b, npad = packing.pack_conv_winograd4(kernel, bias, [(c_a, 32), (c_b, 16)])
    assert wp.shape == (3, 72, 4, 2, 16, 2, 2) and npad == 64 and wp.dtype == np.float32
    xa, xb = rng.normal(size=(8, 12, c_a)).astype(np.float32), rng.normal(size=(8, 12, c_b)).astype(np.float32)
    x = np.zeros((8, 12, 48), np.float32)
    x[..., :c_a], x[..., 32:] = xa, xb
    got = _wino4_emulate(x, wp, b, cout)
    xcat = np.concatenate([xa, xb], -1).astype(np.float64)
    xpad = np.pad(xcat, ((1, 1), (1, 1), (0, 0)))
    want = np.zeros((8, 12, cout))
    for u in range(3):
        for v in range(3):
            want += np.einsum('hwc,co->hwo', xpad[u:u + 8, v:v + 12], kernel[u, v].astype(np.float64))
    want += bias
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    with pytest.raises(ValueError):
        packing.pack_conv_winograd4(kernel, bias, [(c_a, 24), (c_b, 16)])


def test_conditioning_fixture_covers_the_cases_the_gpu_tests_gate_on():
    """tests/golden/conditioning.json (make_conditioning.py): the oracle's own fp32-vs-fp64 agreement for every free-running
    case -- the contractive regime of the north-star tests incl. every element of the seed-3 batch of 8, and the mid regime
    (round 3), where the oracle itself flips on some seeds."""
    import json
    from tf_raft_amd import weights as wm
    with open(os.path.join(ROOT, 'tests', 'golden', 'conditioning.json')) as f:
        fx = json.load(f)
    for b in range(8):
        assert max(fx[f'raft_448x512_seed3_it24_conditioned_batch8_element{b}']['epe32v64']) <= 2e-4
    clean = [s for s in range(5) if max(fx[f'raft_448x512_seed{s}_it24_mid']['epe32v64']) <= 1e-3]
    assert len(clean) >= 4 and 2 not in clean            # seed 2: the oracle flips a tap at iteration 18
    assert max(fx['raft_448x512_seed0_it24_mid']['max_abs_flow']) > 8 * 5      # multi-pixel flow at 1/8 resolution
    w = wm.init_weights('raft', seed=0)
    k = 'update_block/flow_head/conv2/'
    mid, con = wm.condition_weights('raft', w, 'mid'), wm.condition_weights('raft', w)
    np.testing.assert_allclose(mid[k + 'kernel'], w[k + 'kernel'] * np.float32(0.2))
    np.testing.assert_allclose(con[k + 'kernel'], w[k + 'kernel'] * np.float32(0.01))
    assert tuple(mid[k + 'bias']) == (np.float32(0.2), np.float32(-0.15)) and mid['fnet/conv1/kernel'] is w['fnet/conv1/kernel']
    with pytest.raises(ValueError):
        wm.condition_weights('raft', w, 'strong')


def test_dropout_seeds_differ_by_rank_model_seed_and_step():
    """the two dropout masks of a training step are seeded from (model seed, data-parallel rank, optimizer step) --
    ranks see different shards and must not share masks, and a resumed run (optimizer.iterations restored) continues the
    sequence instead of replaying it.  Seeds are even (the second mask uses seed + 1) and fit the C ABI's uint64."""
    from tf_raft_amd.model import _dropout_seed
    seen = set()
    for seed in (0, 1, 7):
        for rank in range(8):
            for step in (1, 2, 3, 1000, 1001):
                s = _dropout_seed(seed, rank, step)
                assert s % 2 == 0 and 0 <= s < 2 ** 63
                seen.add(s)
                seen.add(s + 1)
    assert len(seen) == 2 * 3 * 8 * 5
    assert _dropout_seed(3, 2, 10) == _dropout_seed(3, 2, 10)


def test_pending_results_join_on_every_route_to_their_bytes():
    """tf_raft_amd._dev.DeviceTensor of a pipelined forward call: every way to the data (torch functions and methods, tensors inside
    lists / keyword arguments, data_ptr, as_subclass, numpy, np.asarray) joins the producer first; metadata reads do not.  (The
    stream side is exercised on the GPU: tests/test_gpu_model.py::test_pending_results_join_whichever_stream_touches_them_first.)"""
    from tf_raft_amd import _dev

    class FakePending:
        def __init__(self):
            self.n = 0

        def join(self):
            self.n += 1

    def fresh():
        p = FakePending()
        return _dev.wrap(torch.arange(12, dtype=torch.float32).reshape(3, 4), p), p

    t, p = fresh()
    assert (tuple(t.shape), t.dtype, t.device.type, t.ndim, t.is_cuda, t.requires_grad) == ((3, 4), torch.float32, 'cpu', 2, False, False)
    assert p.n == 0                                            # metadata: no ordering needed
    for touch in (lambda x: x.data_ptr(), lambda x: x[0], lambda x: x + 1, lambda x: torch.cat([x, x]), lambda x: x.sum(),
                  lambda x: torch.add(torch.zeros(3, 4), other=x), lambda x: x.as_subclass(torch.Tensor), lambda x: x.numpy(),
                  lambda x: np.asarray(x), lambda x: x.detach(), lambda x: x.contiguous(), lambda x: _dev.to_device.__wrapped__(x)
                  if hasattr(_dev.to_device, '__wrapped__') else x.clone(), lambda x: _dev.join(x), lambda x: x.__cuda_array_interface__
                  if x.is_cuda else x.tolist()):
        t, p = fresh()
        touch(t)
        assert p.n >= 1, touch
    t, p = fresh()
    view = t[1]                                                # a derived tensor is ordinary: the join has happened already
    assert isinstance(view, _dev.DeviceTensor) and view.__dict__.get('_pending') is None
    plain = _dev.wrap(torch.zeros(2))                          # results of the serial path carry no Pending
    assert plain.__dict__.get('_pending') is None and float((plain + 1).sum()) == 2.0
    assert _dev.join([plain, torch.zeros(1), None]) is not None
    # tensors inside keyword containers at any depth, the dlpack protocol (pickling / deepcopy: next test)
    for touch in (lambda x: torch.cat(tensors=[x, x]), lambda x: torch.stack(tensors=(x, x)), lambda x: torch.from_dlpack(x),
                  lambda x: x.__dlpack__()):
        t, p = fresh()
        touch(t)
        assert p.n >= 1, touch


def test_a_real_pending_pickles_as_none_after_joining():
    """``Pending`` holds a HIP event; a pickled / deep-copied result tensor is read behind a join and carries no Pending."""
    import copy
    import pickle
    from tf_raft_amd import _dev

    class Ev:
        waited = 0

    class P(_dev.Pending):
        def join(self):
            Ev.waited += 1

    t = _dev.wrap(torch.ones(3), P(object(), torch.device('cpu')))
    u = pickle.loads(pickle.dumps(t))
    assert Ev.waited >= 1 and u.__dict__.get('_pending') is None and float(u.sum()) == 3.0
    v = copy.deepcopy(t)
    assert v.__dict__.get('_pending') is None and float(v.sum()) == 3.0
    import io
    buf = io.BytesIO()
    torch.save(t, buf)
    buf.seek(0)
    w = torch.load(buf, weights_only=False)
    assert w.__dict__.get('_pending') is None and float(w.sum()) == 3.0
