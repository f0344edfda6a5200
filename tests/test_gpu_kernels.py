"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle.  GPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.as_tensor(np.asarray(a))


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


# ------------------------------------------------------------------------------------------------
def test_library_is_native_and_loaded():
    from tf_raft_amd import _ffi
    lib = _ffi.load_library()
    assert lib.raft_version() == _ffi.ABI_VERSION
    with open('/proc/self/maps') as f:
        assert 'libraft_hip.so' in f.read()


def test_coords_grid():
    import oracle
    from tf_raft_amd.layers.corr import coords_grid
    got = _np(coords_grid(3, 7, 12))
    ref = oracle.coords_grid(3, 7, 12).numpy()
    np.testing.assert_array_equal(got, ref)
    assert got[0, 2, 5, 0] == 5 and got[0, 2, 5, 1] == 2      # [..., 0] = x, [..., 1] = y


def test_bilinear_sampler_interior_matches_oracle_and_grid_sample(rng):
    """reference tests/layers/test_corr.py:15-27 (sampler == resampler for interior coordinates)."""
    import oracle
    from tf_raft_amd.layers.corr import bilinear_sampler
    n, h, w, r = 4 * 32 * 32, 32, 32, 4
    image = rng.normal(size=(n, h, w, 1)).astype(np.float32)
    cx = rng.uniform(0, w - 1, size=(n, 2 * r + 1, 2 * r + 1)).astype(np.float32)
    cy = rng.uniform(0, h - 1, size=(n, 2 * r + 1, 2 * r + 1)).astype(np.float32)
    coords = np.stack([cx, cy], axis=-1)
    got = _np(bilinear_sampler(image, coords))
    ref = oracle.bilinear_sampler(_t(image), _t(coords)).numpy()
    report('bilinear_sampler', max_abs=float(np.abs(got - ref).max()))
    np.testing.assert_array_equal(got, ref)                    # same unfused arithmetic: bit-exact
    # independent comparator standing in for tfa.image.resampler
    grid = torch.stack([_t(cx) / (w - 1) * 2 - 1, _t(cy) / (h - 1) * 2 - 1], dim=-1)
    gs = torch.nn.functional.grid_sample(_t(image).permute(0, 3, 1, 2), grid, mode='bilinear',
                                         padding_mode='zeros', align_corners=True)
    np.testing.assert_allclose(got[..., 0], gs[:, 0].numpy(), atol=1e-5, rtol=1e-5)


def test_bilinear_sampler_integer_and_out_of_range_are_zero(rng):
    """ceil/floor weights vanish on integer coordinates; clamped samples are integer."""
    import oracle
    from tf_raft_amd.layers.corr import bilinear_sampler
    n, h, w = 6, 5, 7
    image = rng.normal(size=(n, h, w, 1)).astype(np.float32) + 3.0
    coords = np.zeros((n, 3, 4, 2), np.float32)
    coords[..., 0] = rng.uniform(0.1, w - 1.1, size=(n, 3, 4))
    coords[..., 1] = rng.uniform(0.1, h - 1.1, size=(n, 3, 4))
    coords[0, :, :, 0] = 2.0            # integer x
    coords[1, :, :, 1] = 3.0            # integer y
    coords[2, :, :, 0] = -0.75          # out of range (left)
    coords[3, :, :, 0] = w - 1 + 0.25   # out of range (right)
    coords[4, :, :, 1] = h + 10.0       # out of range (bottom)
    got = _np(bilinear_sampler(image, coords))
    ref = oracle.bilinear_sampler(_t(image), _t(coords)).numpy()
    np.testing.assert_array_equal(got, ref)
    assert np.all(got[:5] == 0.0)
    assert np.all(got[5] != 0.0)


@pytest.mark.parametrize('shape', [(2, 8, 12, 256), (1, 56, 64, 256), (2, 9, 13, 128)])
def test_corr_build_matches_oracle_pyramid(rng, shape):
    import oracle
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w, C = shape
    levels = 4 if min(h, w) >= 8 else 3
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    dev = CorrBlock(f1, f2, num_levels=levels, radius=4)
    ref = oracle.CorrBlock(_t(f1), _t(f2), num_levels=levels, radius=4)
    ref64 = oracle.CorrBlock(_t(f1).double(), _t(f2).double(), num_levels=levels, radius=4)
    for l, (g, r, r64) in enumerate(zip(dev.corr_pyramid, ref.corr_pyramid, ref64.corr_pyramid)):
        g = _np(g)
        assert g.shape == tuple(r.shape)
        err = float(np.abs(g - r64.numpy()).max())
        err_ref = float(np.abs(r.numpy() - r64.numpy()).max())
        report(f'corr_build{shape} L{l}', max_abs_vs_f64=err, oracle32_vs_f64=err_ref)
        assert err <= max(4 * err_ref, 2e-5)


@pytest.mark.parametrize('shape', [(2, 56, 64, 256), (1, 44, 60, 64), (3, 128, 128, 32), (2, 44, 60, 128), (5, 56, 64, 256)])
def test_corr_build_xcd_tile_order_is_bitwise_the_plain_grid(rng, shape, raft_opt):
    """The XCD-aware workgroup -> tile mapping of the volume build (one region of the tile plane per XCD; default) covers every
    tile exactly once: the whole pyramid, padding included, is bit for bit what the plain (n, m, batch) grid writes -- at the
    benchmarked map size, a ragged one (partial edge tiles, unequal regions) and a 128 x 128 map (config 4)."""
    from tf_raft_amd.layers.corr import CorrBlock
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    raft_opt.set('RAFT_CORR_XCD', '0')
    plain = CorrBlock(f1, f2, num_levels=4, radius=4)._pyr.clone()
    raft_opt.set('RAFT_CORR_XCD', '1')
    xcd = CorrBlock(f1, f2, num_levels=4, radius=4)._pyr
    assert torch.equal(plain, xcd)


@pytest.mark.parametrize('shape', [(2, 56, 64, 256), (1, 8, 16, 64), (2, 24, 32, 128), (1, 46, 62, 256), (1, 128, 128, 64), (3, 16, 48, 256),
                                   (2, 6, 16, 64), (1, 4, 16, 256)])     # three- and two-level pyramids
def test_corr_build_level1_pooled_in_the_epilogue_matches_the_gemm_columns(rng, shape, raft_opt):
    """Round 6: pyramid level 1 = 2x2 averages of the level-0 ACCUMULATORS (two DPP adds per register in the volume build's epilogue:
    the reference's own order, corr.py:106-114) instead of extra GEMM columns against the pooled fmap2.  Both forms of the same
    linear map: every level of both pyramids within the float64 bound of test_corr_build_matches_oracle_pyramid, the two level-1
    volumes within a few ulps of each other, levels 0 / 2 / 3 and every padding float bit-identical; map sizes with ragged
    query tiles (N < 128), the benchmarked map, the training crop, a 1024 x 1024 frame's map."""
    import oracle
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w, C = shape
    levels = 4 if min(h, w) >= 8 else (3 if min(h, w) >= 6 else 2)
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    raft_opt.set('RAFT_CORR_POOL', '0')
    cols = CorrBlock(f1, f2, num_levels=levels, radius=4)
    raft_opt.set('RAFT_CORR_POOL', '1')
    pooled = CorrBlock(f1, f2, num_levels=levels, radius=4)
    ref64 = oracle.CorrBlock(_t(f1).double(), _t(f2).double(), num_levels=levels, radius=4)
    ref32 = oracle.CorrBlock(_t(f1), _t(f2), num_levels=levels, radius=4)
    off = list(pooled._off)
    a, b = cols._pyr.cpu().numpy(), pooled._pyr.cpu().numpy()
    for l in [k for k in (0, 2, 3) if k < levels]:
        np.testing.assert_array_equal(a[off[l]:off[l + 1]], b[off[l]:off[l + 1]])            # same GEMM columns, padding included
    l1a, l1b = a[off[1]:off[2]], b[off[1]:off[2]]
    assert np.array_equal(l1a == 0, l1b == 0) or int(((l1a == 0) != (l1b == 0)).sum()) <= 2     # the padding of the maps is zero in both
    g1, g0 = _np(pooled.corr_pyramid[1]), _np(cols.corr_pyramid[1])
    r64 = ref64.corr_pyramid[1].numpy()
    err_ref = float(np.abs(ref32.corr_pyramid[1].numpy() - r64).max())
    err_pooled, err_cols = float(np.abs(g1 - r64).max()), float(np.abs(g0 - r64).max())
    report(f'corr_build level 1 {shape}', pooled_vs_f64=err_pooled, gemm_columns_vs_f64=err_cols, oracle32_vs_f64=err_ref,
           pooled_vs_columns=float(np.abs(g1 - g0).max()))
    assert err_pooled <= max(4 * err_ref, 2e-5) and err_cols <= max(4 * err_ref, 2e-5)
    assert float(np.abs(g1 - g0).max()) <= 4e-6 * max(1.0, float(np.abs(r64).max()))


def _device_corr_with_oracle_pyramid(f1, f2, levels, radius):
    """Device CorrBlock whose volume is overwritten with the oracle's values, so the lookup can be
    compared in isolation (bit-exact arithmetic expected)."""
    import oracle
    from tf_raft_amd.layers.corr import CorrBlock
    dev = CorrBlock(f1, f2, num_levels=levels, radius=radius)
    ref = oracle.CorrBlock(_t(f1), _t(f2), num_levels=levels, radius=radius)
    for l in range(levels):
        dev._set_level(l, ref.corr_pyramid[l])
        assert torch.equal(dev.corr_pyramid[l].cpu(), ref.corr_pyramid[l])     # tile / un-tile round trip
    return dev, ref


@pytest.mark.parametrize('radius,shape', [(4, (2, 8, 12, 64)), (3, (1, 16, 24, 32)), (4, (1, 56, 64, 32)),
                                          (4, (1, 14, 20, 32))])
def test_corr_lookup_bit_exact_vs_oracle(rng, radius, shape):
    """Strip kernel with the rows transposed through LDS (4 levels, aligned output: what every model runs); the direct-store
    variant it falls back to otherwise is test_corr_lookup_three_levels."""
    B, h, w, C = shape
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    dev, ref = _device_corr_with_oracle_pyramid(f1, f2, 4, radius)
    import oracle
    grid = oracle.coords_grid(B, h, w).numpy()
    cases = {
        'iteration0_integer_grid': grid,
        'random_flow': grid + rng.normal(scale=3.0, size=grid.shape).astype(np.float32),
        'large_flow_out_of_range': grid + rng.normal(scale=40.0, size=grid.shape).astype(np.float32),
        'half_integer': grid + 0.5,
        'near_border': np.clip(grid + rng.normal(scale=0.01, size=grid.shape), -1, None).astype(np.float32),
    }
    for name, coords in cases.items():
        got = _np(dev.retrieve(coords))
        want = ref.retrieve(_t(coords)).numpy()
        assert got.shape == want.shape == (B, h, w, 4 * (2 * radius + 1) ** 2)
        report(f'corr_lookup r={radius} {name}', max_abs=float(np.abs(got - want).max()),
               nonzero=float((want != 0).mean()))
        np.testing.assert_array_equal(got, want)
    # on the integer grid the level-0 window is identically zero
    got0 = _np(dev.retrieve(grid))
    assert np.all(got0[..., :(2 * radius + 1) ** 2] == 0.0)


def test_corr_lookup_three_levels(rng):
    """num_levels = 3: the strips of the fourth level idle, direct stores."""
    B, h, w, C, r = 1, 8, 12, 32, 4
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    dev, ref = _device_corr_with_oracle_pyramid(f1, f2, 3, r)
    import oracle
    coords = oracle.coords_grid(B, h, w).numpy() + rng.normal(scale=2.0, size=(B, h, w, 2)).astype(np.float32)
    got = _np(dev.retrieve(coords))
    want = ref.retrieve(_t(coords)).numpy()
    assert got.shape == want.shape == (B, h, w, 3 * 81)
    np.testing.assert_array_equal(got, want)


def test_corr_lookup_axis_quirk(rng):
    """window axis 0 (index a) offsets x, axis 1 (index b) offsets y."""
    B, h, w, C, r = 1, 16, 16, 32, 4
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    dev, ref = _device_corr_with_oracle_pyramid(f1, f2, 4, r)
    import oracle
    coords = oracle.coords_grid(B, h, w).numpy() + np.float32(0.25)
    got = _np(dev.retrieve(coords))
    q = 8 * w + 8                                   # query pixel (y=8, x=8)
    img = ref.corr_pyramid[0][q, :, :, 0].numpy()
    a, b = 6, 1                                     # x offset +2, y offset -3
    sx, sy = 8.25 + (a - r), 8.25 + (b - r)
    x0, y0 = int(np.floor(sx)), int(np.floor(sy))
    want = ((1 - (sy - y0)) * (1 - (sx - x0)) * img[y0, x0] + (1 - (sy - y0)) * (sx - x0) * img[y0, x0 + 1]
            + (sy - y0) * (1 - (sx - x0)) * img[y0 + 1, x0] + (sy - y0) * (sx - x0) * img[y0 + 1, x0 + 1])
    assert abs(got[0, 8, 8, a * 9 + b] - want) < 1e-5


@pytest.mark.parametrize('shape,sigma', [((1, 56, 64), 3.0), ((4, 56, 64), 0.3), ((2, 9, 13), 40.0), ((1, 5, 7), 1.0)])
def test_lookup_fused_into_convc1_matches_oracle(rng, shape, sigma):
    """raft_lookup_convc1_f32 = relu(convc1(CorrBlock.retrieve(coords))) in one kernel (reference corr.py:116-152 +
    update.py:91, 98).  Against the float64 oracle (lookup on the same volume, then the 1x1 convolution) at the per-kernel
    2e-5 bound, and against the two-kernel HIP path (bit-exact lookup + direct convolution): same window values, the
    324-long sums only differ in order.  (4, 56, 64) is the benchmarked launch; 9x13 / 5x7 have ragged last workgroups
    (117 and 35 queries, 28 per workgroup) and degenerate pyramid levels."""
    import oracle
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    B, h, w = shape
    C = 32
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    levels = 4 if min(h, w) >= 8 else 3
    if levels != 4:
        pytest.skip('the fused kernel is instantiated for the 4-level pyramid')
    dev, ref = _device_corr_with_oracle_pyramid(f1, f2, 4, 4)
    grid = oracle.coords_grid(B, h, w).numpy()
    coords = (grid + rng.normal(scale=sigma, size=grid.shape)).astype(np.float32)
    kernel = (rng.normal(size=(1, 1, 324, 256)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(256,)).astype(np.float32)
    wp, b, npad = packing.pack_convc1_fused(kernel, bias)
    wp_d, b_d, c_d = _dev.to_device(wp), _dev.to_device(b), _dev.to_device(coords)
    out = torch.full((B, h, w, 256), float('nan'), device=c_d.device)
    check(_dev.lib().raft_lookup_convc1_f32(_dev.ptr(dev._pyr), dev._off, _dev.ptr(c_d), B, h, w, _dev.ptr(wp_d), _dev.ptr(b_d),
                                            npad, 256, _dev.ptr(out), 256, _dev.stream_ptr()), 'lookup_convc1')
    got = _np(out)
    corr64 = ref.retrieve(_t(coords)).double()                      # the oracle's lookup of the SAME fp32 volume
    want = torch.relu(tf_ops.conv2d(corr64, _t(kernel).double(), _t(bias).double())).numpy()
    err = float(np.abs(got - want).max())
    two = _conv_device([(_np(dev.retrieve(coords)), 352)], kernel, bias, act=1)          # lookup kernel + direct 1x1 kernel
    report(f'lookup+convc1 fused {shape} sigma={sigma}', max_abs_vs_f64=err, scale=float(np.abs(want).max()),
           vs_two_kernels=float(np.abs(got - two).max()))
    assert not np.isnan(got).any()
    assert err <= 2e-5 * max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got - two).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize('radius,C', [(4, 256), (3, 128)])
# kernel: blocked MFMA (4 x 8 query blocks, default) / wave per query;  flow: smooth-ish (one bounding box of targets
# per block) / wildly divergent (blocks fall back to one query at a time);  shapes: whole blocks / ragged edges
@pytest.mark.parametrize('block', ['1', '0'])
# (1, 48, 64) at sigma 30: union boxes of more than 96 runs -> the per-query fallback inside the blocked kernel, both (r, C)
@pytest.mark.parametrize('shape,sigma', [((2, 16, 24), 4.0), ((1, 18, 21), 1.0), ((1, 16, 24), 40.0), ((1, 48, 64), 30.0)])
def test_corr_lookup_ondemand_matches_volume(rng, radius, C, shape, sigma, block, raft_opt):
    from tf_raft_amd.layers.corr import CorrBlock
    raft_opt.set('RAFT_ONDEMAND_BLOCK', block)
    B, h, w = shape
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    vol = CorrBlock(f1, f2, 4, radius)
    alt = CorrBlock(f1, f2, 4, radius, alternate=True)
    import oracle
    coords = oracle.coords_grid(B, h, w).numpy() + rng.normal(scale=sigma, size=(B, h, w, 2)).astype(np.float32)
    a, b = _np(vol.retrieve(coords)), _np(alt.retrieve(coords))
    report(f'ondemand r={radius} C={C} {shape} sigma={sigma} block={block}', max_abs=float(np.abs(a - b).max()),
           scale=float(np.abs(a).max()), nonzero=float((a != 0).mean()))
    np.testing.assert_allclose(b, a, atol=2e-5 * max(1.0, float(np.abs(a).max())), rtol=0)
    with pytest.raises(AttributeError):
        alt.corr_pyramid


@pytest.mark.parametrize('shape,C,sigma', [((1, 128, 128), 256, 2.0), ((1, 128, 128), 256, 30.0), ((1, 56, 64), 256, 3.0),
                                           ((2, 19, 27), 128, 1.0)])
def test_corr_lookup_ondemand_matches_oracle(rng, shape, C, sigma):
    """BASELINE config 4 ((1,1024,1024,3) -> 128 x 128 feature maps): the volume-free lookup against the ORACLE's
    stored-volume CorrBlock.retrieve (reference corr.py:116-152 on the 1.07 GB volume of corr.py:154-162), not against
    the HIP volume path.  The on-demand kernel sums the C products in MFMA order, so the bound is the per-kernel 2e-5
    (relative to the correlation scale), not bit-exactness."""
    import oracle
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w = shape
    radius = 4
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    alt = CorrBlock(f1, f2, 4, radius, alternate=True)
    ref = oracle.CorrBlock(_t(f1), _t(f2), 4, radius)
    grid = oracle.coords_grid(B, h, w).numpy()
    cases = {'noisy_flow': grid + rng.normal(scale=sigma, size=grid.shape).astype(np.float32),
             'integer_grid': grid, 'half_integer': grid + np.float32(0.5)}
    for name, coords in cases.items():
        got = _np(alt.retrieve(coords))
        want = ref.retrieve(_t(coords)).numpy()
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got - want).max())
        report(f'ondemand-vs-oracle {shape} C={C} sigma={sigma} {name}', max_abs=err, scale=scale,
               nonzero=float((want != 0).mean()))
        assert got.shape == want.shape
        assert err <= 2e-5 * scale
        # the zero pattern of the sampler (integer / clamped taps) must be reproduced: structural zeros are
        # exact zeros in both; a NON-structural value may cancel to exactly 0 in one summation order only (probability
        # ~1e-7 per element), so a handful of mismatches is tolerated -- never a whole tap row / column
        assert int(((got == 0) != (want == 0)).sum()) <= 4
        if name == 'integer_grid':
            assert np.all(got[..., :(2 * radius + 1) ** 2] == 0.0)       # level 0 on the integer grid: identically 0


@pytest.mark.parametrize('shape', [(2, 7, 9), (1, 6, 8), (3, 2, 1),   # odd width: the last pixel pair is half empty
                                   (1, 56, 64), (1, 128, 128), (2, 46, 62)])   # BASELINE's 448x512 / 1024x1024 maps, the training crop
def test_upsample_convex_matches_oracle(rng, shape):
    from oracle.model import upsample_flow
    from tf_raft_amd import RAFT
    B, h, w = shape
    flow = rng.normal(scale=5.0, size=(B, h, w, 2)).astype(np.float32)
    mask = rng.normal(scale=2.0, size=(B, h, w, 576)).astype(np.float32)
    want = upsample_flow(_t(flow), _t(mask)).numpy()
    got = _np(RAFT.upsample_flow(None, flow, mask))
    assert got.shape == (B, 8 * h, 8 * w, 2)
    report('upsample_convex', max_abs=float(np.abs(got - want).max()))
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)


def test_upflow8_matches_oracle_and_torch(rng):
    import oracle
    from tf_raft_amd.layers.corr import upflow8
    flow = rng.normal(scale=5.0, size=(2, 6, 11, 2)).astype(np.float32)
    got = _np(upflow8(flow))
    want = oracle.upflow8(_t(flow)).numpy()
    report('upflow8', max_abs=float(np.abs(got - want).max()))
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=2e-6)     # |values| ~ 60: a few fp32 ulps
    ti = 8 * torch.nn.functional.interpolate(_t(flow).permute(0, 3, 1, 2), scale_factor=8, mode='bilinear',
                                             align_corners=False).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, ti, atol=1e-4, rtol=1e-5)


# ------------------------------------------------------------------------------------------------
def _conv_device(x_srcs, kernel, bias, act, scale=1.0, nvalid=None):
    """Run raft_conv2d_f32 on NHWC sources [(array, c_pad), ...]."""
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    kh, kw, cin, cout = kernel.shape
    srcs = []
    for arr, cpad in x_srcs:
        B, H, W, c = arr.shape
        buf = np.zeros((B, H, W, cpad), np.float32)
        buf[..., :c] = arr
        srcs.append(_dev.to_device(buf))
    wp, b, npad = packing.pack_conv(kernel, bias, [(a.shape[-1], cp) for a, cp in x_srcs])
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    nvalid = nvalid or cout
    out = torch.full((B, H, W, nvalid), float('nan'), device=srcs[0].device)
    a1 = srcs[1] if len(srcs) > 1 else None
    check(_dev.lib().raft_conv2d_f32(_dev.ptr(srcs[0]), srcs[0].shape[-1], srcs[0].shape[-1],
                                     _dev.ptr(a1) if a1 is not None else None,
                                     a1.shape[-1] if a1 is not None else 0, a1.shape[-1] if a1 is not None else 0,
                                     _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, kh, kw, npad, nvalid, act, scale,
                                     _dev.ptr(out), nvalid, _dev.stream_ptr()), 'conv2d')
    torch.cuda.synchronize()
    return _np(out)


@pytest.mark.parametrize('tile', ['141', '142', '171', '172', '181', '182'])   # conv.hip tile codes
@pytest.mark.parametrize('ksize', [(1, 1), (3, 3), (1, 5), (5, 1)])
def test_conv2d_mfma_matches_oracle(rng, ksize, tile, raft_opt):
    from oracle import tf_ops
    kh, kw = ksize
    B, H, W = 2, 9, 13                      # M = 234: exercises the M tail of every tile
    c_a, c_b, cout = 40, 64, 150            # two sources, first one padded 40 -> 64; N tail 150 -> 192
    xa = rng.normal(size=(B, H, W, c_a)).astype(np.float32)
    xb = rng.normal(size=(B, H, W, c_b)).astype(np.float32)
    kernel = (rng.normal(size=(kh, kw, c_a + c_b, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    raft_opt.set('RAFT_CONV_TILE', tile)
    got = _conv_device([(xa, 64), (xb, 64)], kernel, bias, act=1, scale=0.5)
    x = torch.cat([_t(xa), _t(xb)], dim=-1)
    want = 0.5 * torch.relu(tf_ops.conv2d(x.double(), _t(kernel).double(), _t(bias).double())).numpy()
    err = float(np.abs(got - want).max())
    report(f'conv2d {ksize} tile {tile}', max_abs_vs_f64=err)
    assert not np.isnan(got).any()
    assert err < 2e-5


# kernel variants: channel blocks of 32 / 64 (TNW), pinned weight prefetch on / off (SB), 16 / 32 channels per barrier (CK)
# + K split between two wave sets of a 512-thread workgroup (KS: the single-pair launches), forced on and off
@pytest.mark.parametrize('variant', ['tnw1', 'tnw2', 'tnw1-sb0-ck1', 'tnw1-sb0-ck2', 'tnw1-sb1-ck1', 'tnw2-sb0', 'tnw1-ck2-ks2',
                                     'tnw1-ck4-ks2', 'tnw1-ck2-ks1'])
@pytest.mark.parametrize('shape', [(2, 9, 13), (1, 8, 64), (1, 5, 35)])      # ragged tiles, exact tiles, 2 x-tiles + tail
def test_conv2d_winograd_matches_oracle(rng, shape, variant, raft_opt):
    """Winograd F(2x2, 3x3) kernel (conv_wino.h) against the float64 direct convolution; two sources, N tail."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    for part in variant.split('-'):
        key = {'tnw': 'RAFT_WINO_TNW', 'sb': 'RAFT_WINO_SB', 'ck': 'RAFT_WINO_CK', 'ks': 'RAFT_WINO_KS'}[part.rstrip('0124')]
        raft_opt.set(key, part[-1])
    tnw = variant
    B, H, W = shape
    c_a, c_b, cout = 40, 64, 150
    xa = rng.normal(size=(B, H, W, c_a)).astype(np.float32)
    xb = rng.normal(size=(B, H, W, c_b)).astype(np.float32)
    kernel = (rng.normal(size=(3, 3, c_a + c_b, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    srcs = []
    pad_a = 64 if ('ck2' in variant or 'ck4' in variant) else 48   # 32 / 64 channels per barrier: sources in multiples of that
    for arr, cpad in ((xa, pad_a), (xb, 64)):                  # the winograd kernel walks 16-channel chunks
        buf = np.zeros((B, H, W, cpad), np.float32)
        buf[..., :arr.shape[-1]] = arr
        srcs.append(_dev.to_device(buf))
    # pack_conv pads sources to multiples of 32: pad 40 -> 64 rows, of which the kernel walks the first pad_a
    wp, b, npad = packing.pack_conv_winograd(kernel, bias, [(c_a, 64), (c_b, 64)])
    assert wp.shape == (16, 32, npad, 4)
    if pad_a == 48:
        wp = np.ascontiguousarray(np.concatenate([wp[:, :12], wp[:, 16:]], axis=1))  # drop the 16 all-zero rows 48..63
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    out = torch.full((B, H, W, cout), float('nan'), device=srcs[0].device)
    check(_dev.lib().raft_conv2d_winograd_f32(_dev.ptr(srcs[0]), pad_a, pad_a, _dev.ptr(srcs[1]), 64, 64, _dev.ptr(wp_d),
                                              _dev.ptr(b_d), B, H, W, npad, cout, 1, 0.5, _dev.ptr(out), cout,
                                              _dev.stream_ptr()), 'conv2d_winograd')
    torch.cuda.synchronize()
    got = _np(out)
    x = torch.cat([_t(xa), _t(xb)], dim=-1)
    want = 0.5 * torch.relu(tf_ops.conv2d(x.double(), _t(kernel).double(), _t(bias).double())).numpy()
    direct = _conv_device([(xa, 64), (xb, 64)], kernel, bias, act=1, scale=0.5)
    err, err_direct = float(np.abs(got - want).max()), float(np.abs(direct - want).max())
    report(f'conv2d winograd {shape} tnw {tnw}', max_abs_vs_f64=err, direct_vs_f64=err_direct)
    assert not np.isnan(got).any()
    assert err < 2e-5


@pytest.mark.parametrize('ks', [1, 2])                      # 8 x 64-pixel workgroups / 4 x 64-pixel workgroups with K split in two
@pytest.mark.parametrize('shape', [(2, 9, 13), (1, 8, 64), (1, 5, 70), (2, 16, 128), (1, 56, 64)])   # ragged, exact, 2 x-tiles + tail, big
def test_conv2d_winograd4_matches_oracle(rng, shape, ks, raft_opt):
    """Winograd F(4x4, 3x3) kernel (conv_wino4.h) against the float64 direct convolution; two sources (the first one
    padded 40 -> 48 channels), N tail 150 -> 192.  Its fp32 deviation is ~3x that of the F(2x2, 3x3) kernel (reported
    next to it and to the direct kernel): bound 6e-5 on outputs of magnitude ~10 where the other kernels have 2e-5."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    B, H, W = shape
    c_a, c_b, cout = 40, 64, 150
    raft_opt.set('RAFT_WINO4_KS', str(ks))
    pad_a = 48 if ks == 1 else 64                # the K split pairs up the 16-channel chunks of each source: multiples of 32
    xa = rng.normal(size=(B, H, W, c_a)).astype(np.float32)
    xb = rng.normal(size=(B, H, W, c_b)).astype(np.float32)
    kernel = (rng.normal(size=(3, 3, c_a + c_b, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    srcs = []
    for arr, cpad in ((xa, pad_a), (xb, 64)):
        buf = np.zeros((B, H, W, cpad), np.float32)
        buf[..., :arr.shape[-1]] = arr
        srcs.append(_dev.to_device(buf))
    wp, b, npad = packing.pack_conv_winograd4(kernel, bias, [(c_a, pad_a), (c_b, 64)])
    assert wp.shape == ((pad_a + 64) // 16, 72, 4, npad // 32, 16, 2, 2) and npad == 192
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    out = torch.full((B, H, W, cout), float('nan'), device=srcs[0].device)
    check(_dev.lib().raft_conv2d_winograd4_f32(_dev.ptr(srcs[0]), pad_a, pad_a, _dev.ptr(srcs[1]), 64, 64, _dev.ptr(wp_d),
                                               _dev.ptr(b_d), B, H, W, npad, cout, 1, 0.5, _dev.ptr(out), cout,
                                               _dev.stream_ptr()), 'conv2d_winograd4')
    torch.cuda.synchronize()
    got = _np(out)
    x = torch.cat([_t(xa), _t(xb)], dim=-1)
    want = 0.5 * torch.relu(tf_ops.conv2d(x.double(), _t(kernel).double(), _t(bias).double())).numpy()
    direct = _conv_device([(xa, 64), (xb, 64)], kernel, bias, act=1, scale=0.5)
    err, err_direct = float(np.abs(got - want).max()), float(np.abs(direct - want).max())
    report(f'conv2d winograd F(4x4,3x3) {shape} ks {ks}', max_abs_vs_f64=err, direct_vs_f64=err_direct,
           rms_vs_f64=float(np.sqrt(((got - want) ** 2).mean())), rms_direct=float(np.sqrt(((direct - want) ** 2).mean())))
    assert not np.isnan(got).any()
    if err >= 6e-5:   # locate a structural fault from one run: worst pixel / channel and the error pattern over tile positions
        e = np.abs(got - want)
        bb, yy, xx, nn = np.unravel_index(int(e.argmax()), e.shape)
        pos = np.array([[e[:, i::4, j::4].max() for j in range(4)] for i in range(4)])
        print(f'[wino4] worst at b {bb} y {yy} x {xx} n {nn}: got {got[bb, yy, xx, nn]} want {want[bb, yy, xx, nn]}')
        print('[wino4] max error by position inside the 4x4 tile:\n', pos)
        print('[wino4] max error by channel block of 16:', [float(e[..., k:k + 16].max()) for k in range(0, cout, 16)])
        print('[wino4] max error by row:', [float(e[:, y].max()) for y in range(H)])
        print('[wino4] max error by column (first 72):', [round(float(e[:, :, x].max()), 5) for x in range(min(W, 72))])
    assert err < 6e-5


def test_conv2d_winograd4_linear_residual_and_rejections(rng):
    """The linear epilogue with a scale, one source, channel counts that are exact multiples; argument rejections."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    B, H, W, cin, cout = 1, 12, 20, 32, 64
    x = rng.normal(size=(B, H, W, cin)).astype(np.float32)
    kernel = (rng.normal(size=(3, 3, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    wp, b, npad = packing.pack_conv_winograd4(kernel, bias)
    xd, wp_d, b_d = _dev.to_device(x), _dev.to_device(wp), _dev.to_device(b)
    out = torch.full((B, H, W, cout), float('nan'), device=xd.device)
    check(_dev.lib().raft_conv2d_winograd4_f32(_dev.ptr(xd), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, npad,
                                               cout, 0, 2.0, _dev.ptr(out), cout, _dev.stream_ptr()), 'conv2d_winograd4')
    torch.cuda.synchronize()
    want = 2.0 * tf_ops.conv2d(_t(x).double(), _t(kernel).double(), _t(bias).double()).numpy()
    err = float(np.abs(_np(out) - want).max())
    report('conv2d winograd F(4x4,3x3) linear', max_abs_vs_f64=err)
    assert err < 6e-5
    rc = _dev.lib().raft_conv2d_winograd4_f32(_dev.ptr(xd), cin, 24, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, npad,
                                              cout, 0, 1.0, _dev.ptr(out), cout, _dev.stream_ptr())
    assert rc < 0                                                    # 24 channels: not a multiple of 16


@pytest.mark.parametrize('tnw', ['1', '2', '1-ck2', '2-ck2', '2-ck2-tm1', '1-tm1', '2-tm2'])   # TNW, CK, TM variants
@pytest.mark.parametrize('ksize', [(1, 5), (5, 1)])
@pytest.mark.parametrize('shape', [(2, 9, 13), (1, 8, 64), (1, 21, 35)])
def test_conv1d_winograd_matches_oracle(rng, shape, ksize, tnw, raft_opt):
    """1-D Winograd F(2, 5) kernel (conv_wino1d.h) against the float64 direct convolution; two sources, N tail."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    raft_opt.set('RAFT_WINO_TNW', tnw[0])
    raft_opt.set('RAFT_WINO_CK', '2' if 'ck2' in tnw else '1')
    if 'tm' in tnw:
        raft_opt.set('RAFT_WINO1D_TM', tnw[-1])
    kh, kw = ksize
    B, H, W = shape
    c_a, c_b, cout = (64 if 'ck2' in tnw else 48), 64, 150          # 32 channels per barrier: sources in multiples of 32
    xa = rng.normal(size=(B, H, W, c_a)).astype(np.float32)
    xb = rng.normal(size=(B, H, W, c_b)).astype(np.float32)
    kernel = (rng.normal(size=(kh, kw, c_a + c_b, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    sa, sb = _dev.to_device(xa), _dev.to_device(xb)
    wp, b, npad = packing.pack_conv_winograd1d(kernel, bias, [(c_a, 64), (c_b, 64)])
    assert wp.shape == (6, 32, npad, 4)
    if c_a == 48:
        wp = np.ascontiguousarray(np.concatenate([wp[:, :12], wp[:, 16:]], axis=1))  # drop the all-zero rows 48..63
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    out = torch.full((B, H, W, cout), float('nan'), device=sa.device)
    check(_dev.lib().raft_conv1d_winograd_f32(_dev.ptr(sa), c_a, c_a, _dev.ptr(sb), c_b, c_b, _dev.ptr(wp_d), _dev.ptr(b_d),
                                              B, H, W, kh, kw, npad, cout, 1, 0.5, _dev.ptr(out), cout, _dev.stream_ptr()),
          'conv1d_winograd')
    torch.cuda.synchronize()
    got = _np(out)
    x = torch.cat([_t(xa), _t(xb)], dim=-1)
    want = 0.5 * torch.relu(tf_ops.conv2d(x.double(), _t(kernel).double(), _t(bias).double())).numpy()
    direct = _conv_device([(xa, 64), (xb, 64)], kernel, bias, act=1, scale=0.5)
    err, err_direct = float(np.abs(got - want).max()), float(np.abs(direct - want).max())
    report(f'conv1d winograd {ksize} {shape} tnw {tnw}', max_abs_vs_f64=err, direct_vs_f64=err_direct)
    assert not np.isnan(got).any()
    assert err < 2e-5


@pytest.mark.parametrize('tnw', ['1', '2'])
@pytest.mark.parametrize('ksize', [(1, 5), (5, 1)])
@pytest.mark.parametrize('shape', [(2, 9, 13), (1, 8, 64), (1, 21, 35), (1, 3, 131), (1, 67, 5)])
def test_conv1d_winograd4_matches_oracle(rng, shape, ksize, tnw, raft_opt):
    """1-D Winograd F(4, 5) kernel (conv_wino1d.h, MO = 4) against the float64 direct convolution; two sources, N tail,
    tiles cut by the right / bottom border."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    raft_opt.set('RAFT_WINO_TNW', tnw)
    kh, kw = ksize
    B, H, W = shape
    c_a, c_b, cout = 32, 64, 150
    xa = rng.normal(size=(B, H, W, c_a)).astype(np.float32)
    xb = rng.normal(size=(B, H, W, c_b)).astype(np.float32)
    kernel = (rng.normal(size=(kh, kw, c_a + c_b, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    sa, sb = _dev.to_device(xa), _dev.to_device(xb)
    wp, b, npad = packing.pack_conv_winograd1d(kernel, bias, [(c_a, 32), (c_b, 64)], m=4)
    assert wp.shape == (8, 24, npad, 4)
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    out = torch.full((B, H, W, cout), float('nan'), device=sa.device)
    check(_dev.lib().raft_conv1d_winograd4_f32(_dev.ptr(sa), c_a, c_a, _dev.ptr(sb), c_b, c_b, _dev.ptr(wp_d), _dev.ptr(b_d),
                                               B, H, W, kh, kw, npad, cout, 1, 0.5, _dev.ptr(out), cout, _dev.stream_ptr()),
          'conv1d_winograd4')
    torch.cuda.synchronize()
    got = _np(out)
    x = torch.cat([_t(xa), _t(xb)], dim=-1)
    want = 0.5 * torch.relu(tf_ops.conv2d(x.double(), _t(kernel).double(), _t(bias).double())).numpy()
    direct = _conv_device([(xa, 32), (xb, 64)], kernel, bias, act=1, scale=0.5)
    err, err_direct = float(np.abs(got - want).max()), float(np.abs(direct - want).max())
    report(f'conv1d winograd F(4,5) {ksize} {shape} tnw {tnw}', max_abs_vs_f64=err, direct_vs_f64=err_direct)
    assert not np.isnan(got).any()
    assert err < 2e-5


def test_conv1d_winograd4_rejects_16_channel_sources(rng):
    """F(4, 5) stages 32 channels per barrier: sources that are not multiples of 32 are refused, not mis-computed."""
    from tf_raft_amd import _dev, packing
    x = _dev.to_device(rng.normal(size=(1, 4, 8, 48)).astype(np.float32))
    kernel = rng.normal(size=(1, 5, 48, 32)).astype(np.float32)
    wp, b, npad = packing.pack_conv_winograd1d(kernel, np.zeros(32, np.float32), [(48, 64)], m=4)
    wp_d, b_d = _dev.to_device(np.ascontiguousarray(wp[:, :12])), _dev.to_device(b)
    out = torch.zeros((1, 4, 8, 32), device=x.device)
    rc = _dev.lib().raft_conv1d_winograd4_f32(_dev.ptr(x), 48, 48, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), 1, 4, 8, 1, 5,
                                              npad, 32, 0, 1.0, _dev.ptr(out), 32, _dev.stream_ptr())
    torch.cuda.synchronize()
    assert rc != 0


def test_winograd_kernels_on_random_small_shapes(rng):
    """Shapes that do not fill a single tile, single rows / columns, odd sizes: both Winograd kernels and the direct
    halo kernel against the float64 oracle convolution."""
    from oracle import tf_ops
    from tf_raft_amd import _dev, packing
    from tf_raft_amd._ffi import check
    lib = _dev.lib()
    shapes = [(1, 1, 1), (1, 1, 7), (1, 2, 33), (1, 3, 2), (2, 4, 32), (1, 5, 31), (3, 7, 3), (1, 9, 65), (1, 17, 16),
              (2, 1, 40), (1, 12, 34), (1, 33, 5)]
    for (B, H, W) in shapes:
        cin, cout = int(rng.choice([16, 32, 48])), int(rng.choice([7, 64, 96]))
        x = rng.normal(size=(B, H, W, cin)).astype(np.float32)
        xd = _dev.to_device(x)
        for kh, kw in ((3, 3), (1, 5), (5, 1)):
            kernel = (rng.normal(size=(kh, kw, cin, cout)) * 0.1).astype(np.float32)
            bias = rng.normal(size=(cout,)).astype(np.float32)
            want = tf_ops.conv2d(_t(x).double(), _t(kernel).double(), _t(bias).double()).numpy()
            direct = _conv_device([(x, 32 * ((cin + 31) // 32))], kernel, bias, act=0)
            np.testing.assert_allclose(direct, want, atol=3e-5, rtol=0, err_msg='direct ' + str((B, H, W, kh, kw, cin, cout)))
            out = torch.full((B, H, W, cout), float('nan'), device=xd.device)
            pack = packing.pack_conv_winograd if kh == 3 else packing.pack_conv_winograd1d
            wp, b, npad = pack(kernel, bias, [(cin, 32 * ((cin + 31) // 32))])
            wp_d, b_d = _dev.to_device(np.ascontiguousarray(wp[:, :cin // 4])), _dev.to_device(b)   # keep alive over the launch
            if kh == 3:
                check(lib.raft_conv2d_winograd_f32(_dev.ptr(xd), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W,
                                                   npad, cout, 0, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()), 'conv2d_winograd')
            else:
                check(lib.raft_conv1d_winograd_f32(_dev.ptr(xd), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W,
                                                   kh, kw, npad, cout, 0, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()),
                      'conv1d_winograd')
            torch.cuda.synchronize()
            got = _np(out)
            assert not np.isnan(got).any(), (B, H, W, kh, kw, cin, cout)
            np.testing.assert_allclose(got, want, atol=3e-5, rtol=0, err_msg=str((B, H, W, kh, kw, cin, cout)))
            if kh != 3 and cin % 32 == 0:   # F(4, 5)
                out4 = torch.full((B, H, W, cout), float('nan'), device=xd.device)
                wp4, b4, npad4 = packing.pack_conv_winograd1d(kernel, bias, [(cin, cin)], m=4)
                wp4_d, b4_d = _dev.to_device(wp4), _dev.to_device(b4)
                check(lib.raft_conv1d_winograd4_f32(_dev.ptr(xd), cin, cin, None, 0, 0, _dev.ptr(wp4_d), _dev.ptr(b4_d), B, H, W,
                                                    kh, kw, npad4, cout, 0, 1.0, _dev.ptr(out4), cout, _dev.stream_ptr()),
                      'conv1d_winograd4')
                torch.cuda.synchronize()
                got4 = _np(out4)
                assert not np.isnan(got4).any(), (B, H, W, kh, kw, cin, cout)
                np.testing.assert_allclose(got4, want, atol=3e-5, rtol=0, err_msg='F(4,5) ' + str((B, H, W, kh, kw, cin, cout)))


def test_basic_update_block_winograd_gru_matches_direct(rng, raft_opt):
    """RAFT_GRU_WINO=15 / RAFT_GRU_WINO4=15: the four per-iteration SepConvGRU convolutions on the F(2, 5) and F(4, 5)
    kernels (gate epilogues + context) against the direct kernels."""
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import BasicUpdateBlock
    wts = wm.init_weights('raft', seed=3, perturb=True)
    for shape in ((1, 56, 64), (2, 9, 13)):
        net, inp, corr, flow = _update_inputs(rng, 'raft', *shape)
        blk = BasicUpdateBlock(filters=128, weights=wts)
        raft_opt.set('RAFT_GRU_WINO', '0')
        raft_opt.set('RAFT_GRU_WINO4', '0')
        dn, dm, dd = [_np(t) for t in blk([net, inp, corr, flow])]
        raft_opt.set('RAFT_GRU_WINO', '15')
        wn, wmk, wd = [_np(t) for t in blk([net, inp, corr, flow])]
        report(f'update block F(2,5) GRU vs direct {shape}', net=float(np.abs(wn - dn).max()),
               mask=float(np.abs(wmk - dm).max()), delta=float(np.abs(wd - dd).max()))
        assert np.abs(wn - dn).max() < 2e-5 and np.abs(wmk - dm).max() < 5e-5 and np.abs(wd - dd).max() < 5e-5
        assert np.abs(wn - dn).max() > 0
        raft_opt.set('RAFT_GRU_WINO4', '15')
        vn, vmk, vd = [_np(t) for t in blk([net, inp, corr, flow])]
        report(f'update block F(4,5) GRU vs direct {shape}', net=float(np.abs(vn - dn).max()),
               mask=float(np.abs(vmk - dm).max()), delta=float(np.abs(vd - dd).max()))
        assert np.abs(vn - dn).max() < 2e-5 and np.abs(vmk - dm).max() < 5e-5 and np.abs(vd - dd).max() < 5e-5
        assert np.abs(vn - dn).max() > 0 and np.abs(vn - wn).max() > 0


def test_small_update_block_winograd_layers_match_direct(rng, raft_opt):
    """RAFT_SMALL_WINO=15: conv, the 3x3 ConvGRU (gate epilogues of conv_wino.h) and flow_head.conv1 of SmallRAFT."""
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import SmallUpdateBlock
    wts = wm.init_weights('small', seed=3, perturb=True)
    for shape in ((1, 56, 64), (2, 9, 13)):
        net, inp, corr, flow = _update_inputs(rng, 'small', *shape)
        blk = SmallUpdateBlock(filters=96, weights=wts)
        raft_opt.set('RAFT_SMALL_WINO', '0')
        dn, _, dd = blk([net, inp, corr, flow])
        dn, dd = _np(dn), _np(dd)
        raft_opt.set('RAFT_SMALL_WINO', '15')
        wn, _, wd = blk([net, inp, corr, flow])
        wn, wd = _np(wn), _np(wd)
        report(f'small update block winograd vs direct {shape}', net=float(np.abs(wn - dn).max()),
               delta=float(np.abs(wd - dd).max()))
        assert np.abs(wn - dn).max() < 2e-5 and np.abs(wd - dd).max() < 5e-5
        assert np.abs(wn - dn).max() > 0


def test_basic_update_block_winograd_layers_match_direct(rng, raft_opt):
    """RAFT_CONV_WINO=15: all four 3x3 layers of the update block on the winograd kernel."""
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import BasicUpdateBlock
    wts = wm.init_weights('raft', seed=3, perturb=True)
    net, inp, corr, flow = _update_inputs(rng, 'raft', 1, 56, 64)
    blk = BasicUpdateBlock(filters=128, weights=wts)
    raft_opt.set('RAFT_CONV_WINO', '0')
    dn, dm, dd = [_np(t) for t in blk([net, inp, corr, flow])]
    raft_opt.set('RAFT_CONV_WINO', '15')
    wn, wmk, wd = [_np(t) for t in blk([net, inp, corr, flow])]
    report('update block winograd vs direct', net=float(np.abs(wn - dn).max()), mask=float(np.abs(wmk - dm).max()),
           delta=float(np.abs(wd - dd).max()))
    assert np.abs(wn - dn).max() < 2e-5 and np.abs(wmk - dm).max() < 5e-5 and np.abs(wd - dd).max() < 5e-5
    assert np.abs(wmk - dm).max() > 0                          # the switch really changed the kernels


def test_conv2d_rejects_bad_arguments():
    from tf_raft_amd import _dev
    lib = _dev.lib()
    x = torch.zeros((1, 4, 4, 32), device='cuda')
    w = torch.zeros((9 * 8 * 64 * 4,), device='cuda')
    b = torch.zeros((64,), device='cuda')
    o = torch.zeros((1, 4, 4, 64), device='cuda')
    args = lambda **kw: [kw.get('a0', _dev.ptr(x)), 32, kw.get('c0', 32), None, 0, 0, _dev.ptr(w), _dev.ptr(b),   # noqa: E731
                         1, 4, 4, kw.get('kh', 3), 3, 64, 64, 0, 1.0, _dev.ptr(o), 64, None]
    assert lib.raft_conv2d_f32(*args()) == 0
    assert lib.raft_conv2d_f32(*args(a0=None)) == -1          # RAFT_E_NULL
    assert lib.raft_conv2d_f32(*args(c0=20)) == -3            # RAFT_E_UNSUPPORTED (not a multiple of 32)
    assert lib.raft_conv2d_f32(*args(kh=7)) == -3             # kernel size not instantiated
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
def _update_inputs(rng, variant, B, h, w):
    from tf_raft_amd.layers.update import _GEOM
    g = _GEOM[variant]
    net = np.tanh(rng.normal(size=(B, h, w, g['hdim']))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, g['cdim'])), 0).astype(np.float32)
    corr = rng.normal(size=(B, h, w, g['corr_used'])).astype(np.float32)
    flow = rng.normal(scale=3.0, size=(B, h, w, 2)).astype(np.float32)
    return net, inp, corr, flow


@pytest.mark.parametrize('shape', [(2, 8, 12), (1, 56, 64)])
def test_basic_update_block_matches_oracle(rng, shape):
    from oracle.layers import W, basic_update_block
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import BasicUpdateBlock
    B, h, w = shape
    wts = wm.init_weights('raft', seed=3, perturb=True)
    net, inp, corr, flow = _update_inputs(rng, 'raft', B, h, w)
    blk = BasicUpdateBlock(filters=128, weights=wts)
    gn, gm, gd = blk([net, inp, corr, flow])
    W64 = W(wts, torch.float64)
    rn, rm, rd = basic_update_block(W64, 'update_block', *[_t(a).double() for a in (net, inp, corr, flow)])
    W32 = W(wts, torch.float32)
    on, om, od = basic_update_block(W32, 'update_block', *[_t(a) for a in (net, inp, corr, flow)])
    for name, g, r, o, tol in (('net', gn, rn, on, 2e-5), ('mask', gm, rm, om, 5e-5), ('delta', gd, rd, od, 5e-5)):
        err = float(np.abs(_np(g) - r.numpy()).max())
        err_o = float(np.abs(o.numpy() - r.numpy()).max())
        report(f'basic_update{shape} {name}', hip_vs_f64=err, oracle32_vs_f64=err_o)
        assert err < max(tol, 4 * err_o)


@pytest.mark.parametrize('shape', [(2, 8, 12), (1, 32, 32)])
def test_small_update_block_matches_oracle(rng, shape):
    from oracle.layers import W, small_update_block
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import SmallUpdateBlock
    B, h, w = shape
    wts = wm.init_weights('small', seed=4, perturb=True)
    net, inp, corr, flow = _update_inputs(rng, 'small', B, h, w)
    blk = SmallUpdateBlock(filters=96, weights=wts)
    gn, gm, gd = blk([net, inp, corr, flow])
    assert gm is None
    W64 = W(wts, torch.float64)
    rn, _, rd = small_update_block(W64, 'update_block', *[_t(a).double() for a in (net, inp, corr, flow)])
    for name, g, r, tol in (('net', gn, rn, 2e-5), ('delta', gd, rd, 5e-5)):
        err = float(np.abs(_np(g) - r.numpy()).max())
        report(f'small_update{shape} {name}', hip_vs_f64=err)
        assert err < tol


@pytest.mark.parametrize('variant', ['raft', 'small'])
@pytest.mark.parametrize('tile', ['141', '142', '171', '172', '181', '182'])
def test_update_blocks_every_conv_tile(rng, variant, tile, raft_opt):
    """GRU / relu / linear epilogues of every instantiated tile (forced through RAFT_CONV_TILE where the
    tile divides the layer's npad), M = 2*9*13 = 234 pixels: M tails of the 64/112/128-row tiles."""
    from oracle.layers import W, basic_update_block, small_update_block
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.update import BasicUpdateBlock, SmallUpdateBlock
    B, h, w = 2, 9, 13
    wts = wm.init_weights(variant, seed=6, perturb=True)
    net, inp, corr, flow = _update_inputs(rng, variant, B, h, w)
    blk = (BasicUpdateBlock(filters=128, weights=wts) if variant == 'raft' else SmallUpdateBlock(filters=96, weights=wts))
    raft_opt.set('RAFT_CONV_TILE', tile)
    gn, gm, gd = blk([net, inp, corr, flow])
    torch.cuda.synchronize()
    fn = basic_update_block if variant == 'raft' else small_update_block
    rn, rm, rd = fn(W(wts, torch.float64), 'update_block', *[_t(a).double() for a in (net, inp, corr, flow)])
    errs = dict(net=float(np.abs(_np(gn) - rn.numpy()).max()), delta=float(np.abs(_np(gd) - rd.numpy()).max()))
    if gm is not None:
        errs['mask'] = float(np.abs(_np(gm) - rm.numpy()).max())
    report(f'update {variant} tile {tile}', **errs)
    assert errs['net'] < 2e-5 and errs['delta'] < 5e-5 and errs.get('mask', 0.0) < 5e-5


def test_update_block_rejects_wrong_shapes(rng):
    from tf_raft_amd.layers.update import BasicUpdateBlock
    blk = BasicUpdateBlock(filters=128)
    net, inp, corr, flow = _update_inputs(rng, 'raft', 1, 8, 8)
    with pytest.raises(ValueError):
        blk([net, inp, corr[..., :100], flow])
    with pytest.raises(ValueError):
        BasicUpdateBlock(filters=64)


# ------------------------------------------------------------------------------------------------
def test_c_abi_from_two_host_threads(rng):
    """the C ABI is re-entrant and thread-safe -- calls only enqueue on the stream they are given and the
    library keeps no per-call state.  Two host threads (ctypes drops the GIL inside a call), each on a stream of its own with
    buffers of its own, run volume build -> pyramid lookup -> fused lookup + convc1 -> convex upsampling and a whole RAFT forward
    concurrently, 12 rounds each; one of the threads launches under a thread-local shape hint (raft_set_thread_concurrency),
    which must not leak into the other.  Every result must equal, bit for bit, what the same thread's work returns when it runs
    alone.  (The host side of the ABI -- argument validation, the option table -- runs under ASan / TSan in
    tests/test_abi_sanitizers.py.)"""
    import threading
    import tf_raft_amd
    from tf_raft_amd import _dev, _ffi, packing
    from tf_raft_amd import weights as wm
    from tf_raft_amd._ffi import check
    from tf_raft_amd.layers.corr import CorrBlock
    lib = _dev.lib()
    B, h, w = 2, 24, 32
    wts = wm.init_weights('raft', seed=5, perturb=True)

    def make_work(seed, hint):
        r = np.random.default_rng(seed)
        f1 = _dev.to_device(r.normal(size=(B, h, w, 256)).astype(np.float32))
        f2 = _dev.to_device(r.normal(size=(B, h, w, 256)).astype(np.float32))
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
        coords = _dev.to_device((np.stack([xs, ys], -1)[None].repeat(B, 0) + r.normal(scale=3.0, size=(B, h, w, 2))).astype(np.float32))
        flow = _dev.to_device(r.normal(scale=4.0, size=(B, h, w, 2)).astype(np.float32))
        mask = _dev.to_device(r.normal(size=(B, h, w, 576)).astype(np.float32))
        kc1 = (r.normal(size=(1, 1, 324, 256)) * 0.05).astype(np.float32)
        wp, bias, npad = packing.pack_convc1_fused(kc1, r.normal(size=256).astype(np.float32))
        wp_d, bias_d = _dev.to_device(wp), _dev.to_device(bias)
        img1 = _dev.to_device(r.uniform(0, 255, size=(1, 128, 160, 3)).astype(np.float32))
        img2 = _dev.to_device(r.uniform(0, 255, size=(1, 128, 160, 3)).astype(np.float32))
        model = tf_raft_amd.RAFT(weights=wts, iters_pred=4, pipeline=False, overlap=False, loop_concurrency=hint)
        stream = torch.cuda.Stream()

        def once():
            with torch.cuda.stream(stream), _ffi.thread_concurrency(hint):
                corr = CorrBlock(f1, f2, 4, 4)
                look = torch.empty((B, h, w, 352), device='cuda')
                corr.retrieve(coords, out=look, ld_out=352)
                cor1 = torch.empty((B, h, w, 256), device='cuda')
                check(lib.raft_lookup_convc1_f32(_dev.ptr(corr._pyr), corr._off, _dev.ptr(coords), B, h, w, _dev.ptr(wp_d), _dev.ptr(bias_d),
                                                 npad, 256, _dev.ptr(cor1), 256, _dev.stream_ptr()), 'lookup_convc1')
                up = torch.empty((B, 8 * h, 8 * w, 2), device='cuda')
                check(lib.raft_upsample_convex_f32(_dev.ptr(flow), _dev.ptr(mask), B, h, w, _dev.ptr(up), _dev.stream_ptr()), 'upsample')
                pred = model([img1, img2])[-1].as_subclass(torch.Tensor)
                outs = [look[..., :324].clone(), cor1, up, pred.clone()]
            stream.synchronize()
            return [o.cpu().numpy() for o in outs]
        return once

    works = [make_work(11, 1), make_work(12, 3)]
    alone = [wk() for wk in works]                       # each thread's work run alone (and once more: deterministic)
    for wk, ref in zip(works, alone):
        for a, b in zip(wk(), ref):
            np.testing.assert_array_equal(a, b)
    errors, rounds = [], 12
    barrier = threading.Barrier(2)

    def run(k):
        try:
            barrier.wait()
            for _ in range(rounds):
                for a, b in zip(works[k](), alone[k]):
                    np.testing.assert_array_equal(a, b)
            assert lib.raft_set_thread_concurrency(1) == 1          # the other thread's hint never became this thread's
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)[:500]))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    report('C ABI from two host threads', rounds_per_thread=rounds, results_compared_bitwise=2 * rounds * 4)
