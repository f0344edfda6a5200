"""Pins the CPU oracle (oracle/) -- against the reference's own known-answer vectors where they
exist (tests/golden/reference_known_answers.json) and against independent torch / NumPy
implementations of each TensorFlow op elsewhere.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import tf_ops
from oracle.layers import W, basic_update_block, encoder, sep_conv_gru
from oracle.model import upsample_flow
from tf_raft_amd import weights as wm

from conftest import GOLDEN


@pytest.fixture(scope='module')
def known():
    with open(os.path.join(GOLDEN, 'reference_known_answers.json')) as f:
        return json.load(f)


# ---------------------------------------------------------------- reference tests/test_model.py:14-41
def test_extract_patches_and_depth_to_space_orderings(known):
    k = known['upsample']
    flow = torch.tensor(k['flow_3x3x2'], dtype=torch.float32)[None]
    unfold = tf_ops.extract_patches_valid(flow, 2)
    np.testing.assert_allclose(unfold[0].numpy(), np.array(k['unfold_2x2_valid_2x2x8']))
    up = tf_ops.depth_to_space(unfold, 2)
    np.testing.assert_allclose(up[0].numpy(), np.array(k['depth_to_space_2_4x4x2']))


def test_extract_patches_same_is_zero_padded_and_matches_torch_unfold(rng):
    x = torch.as_tensor(rng.normal(size=(2, 5, 6, 2)).astype(np.float32))
    got = tf_ops.extract_patches_same(x, 3)                        # depth order (ky, kx, ch)
    unf = F.unfold(x.permute(0, 3, 1, 2), 3, padding=1)             # (B, ch*9, L) ordered (ch, ky, kx)
    unf = unf.reshape(2, 2, 9, 5, 6).permute(0, 3, 4, 2, 1).reshape(2, 5, 6, 18)
    np.testing.assert_array_equal(got.numpy(), unf.numpy())
    assert float(got[0, 0, 0, 0]) == 0.0 and float(got[0, 0, 0, 1]) == 0.0   # top-left tap is padding


# ---------------------------------------------------------------- reference tests/losses/test_losses.py
def _loss_fixture(known):
    k = known['losses']
    flow_gt = np.array(k['flow_gt_3x3x2_plus_0p1'], dtype=np.float64) - k['flow_gt_minus']
    valid = np.array(k['valid_3x3'])
    preds = [np.zeros_like(flow_gt)[None] for _ in range(k['n_predictions'])]
    return flow_gt[None], valid[None], preds, k


def test_sequence_loss_known_answer(known):
    flow_gt, valid, preds, k = _loss_fixture(known)
    want = 0.0
    for i, p in enumerate(preds):
        want += k['gamma'] ** (len(preds) - i - 1) * np.mean(valid[..., None] * np.abs(p - flow_gt))
    got = oracle.sequence_loss((flow_gt, valid), preds, gamma=k['gamma'])
    np.testing.assert_almost_equal(got, want, decimal=6)


def test_end_point_error_known_answer(known):
    flow_gt, valid, preds, k = _loss_fixture(known)
    info = oracle.end_point_error([flow_gt, valid], preds[-1])
    want = np.mean(np.sqrt((np.arange(1, 9) - 0.1) ** 2))
    np.testing.assert_almost_equal(info['epe'], want, decimal=2)
    for u in ('u1', 'u3', 'u5'):
        np.testing.assert_almost_equal(info[u], k[u], decimal=2)


# ---------------------------------------------------------------- reference tests/layers/test_corr.py:15-27
def test_bilinear_sampler_equals_standard_bilinear_on_interior_coords(rng):
    n, h, w, r = 4 * 32 * 32, 32, 32, 4
    image = torch.as_tensor(rng.normal(size=(n, h, w, 1)).astype(np.float32))
    cx = torch.as_tensor(rng.uniform(0, w - 1, size=(n, 2 * r + 1, 2 * r + 1)).astype(np.float32))
    cy = torch.as_tensor(rng.uniform(0, h - 1, size=(n, 2 * r + 1, 2 * r + 1)).astype(np.float32))
    got = oracle.bilinear_sampler(image, torch.stack([cx, cy], dim=-1))
    grid = torch.stack([cx / (w - 1) * 2 - 1, cy / (h - 1) * 2 - 1], dim=-1)
    want = F.grid_sample(image.permute(0, 3, 1, 2), grid, mode='bilinear', padding_mode='zeros',
                         align_corners=True)
    np.testing.assert_allclose(got[..., 0].numpy(), want[:, 0].numpy(), atol=1e-5, rtol=1e-5)


def test_bilinear_sampler_integer_and_out_of_range_coordinates_give_zero(rng):
    """SURVEY F4 (reference corr.py:41-48, 57-60): weights are ceil(g)-g and g-floor(g) after clamping."""
    image = torch.as_tensor(rng.normal(size=(1, 6, 7, 1)).astype(np.float32)) + 5.0
    pts = [(2.0, 1.5), (2.5, 3.0), (-1.25, 2.5), (6.75, 2.5), (3.5, -0.5), (3.5, 5.5), (0.0, 0.0), (6.0, 5.0)]
    coords = torch.tensor(pts, dtype=torch.float32).reshape(1, 2, 4, 2)
    out = oracle.bilinear_sampler(image, coords)
    assert torch.all(out == 0)
    inside = oracle.bilinear_sampler(image, torch.tensor([[[[2.5, 1.5]]]]))
    want = image[0, 1:3, 2:4, 0].mean()
    np.testing.assert_allclose(float(inside), float(want), rtol=1e-6)


def test_coords_grid_is_xy(rng):
    g = oracle.coords_grid(2, 3, 5)
    assert g.shape == (2, 3, 5, 2)
    assert float(g[1, 2, 4, 0]) == 4.0 and float(g[1, 2, 4, 1]) == 2.0


# ---------------------------------------------------------------- TF op semantics vs independent code
def test_same_padding_is_tensorflow_asymmetric():
    assert tf_ops.same_padding(448, 7, 2) == (2, 3)
    assert tf_ops.same_padding(224, 3, 2) == (0, 1)
    assert tf_ops.same_padding(56, 3, 1) == (1, 1)
    assert tf_ops.same_padding(56, 5, 1) == (2, 2)
    assert tf_ops.same_padding(7, 3, 2) == (1, 1)                  # odd input: out = 4, total 2


def test_conv2d_matches_explicit_loops(rng):
    x = rng.normal(size=(1, 6, 7, 3)).astype(np.float32)
    k = rng.normal(size=(3, 3, 3, 4)).astype(np.float32)
    b = rng.normal(size=(4,)).astype(np.float32)
    for stride in (1, 2):
        got = tf_ops.conv2d(torch.as_tensor(x), torch.as_tensor(k), torch.as_tensor(b), stride).numpy()
        pt, pb = tf_ops.same_padding(6, 3, stride)
        pl, pr = tf_ops.same_padding(7, 3, stride)
        xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        ho, wo = -(-6 // stride), -(-7 // stride)
        want = np.zeros((1, ho, wo, 4), np.float64)
        for y in range(ho):
            for xx in range(wo):
                patch = xp[0, y * stride:y * stride + 3, xx * stride:xx * stride + 3, :]
                want[0, y, xx] = np.tensordot(patch, k, axes=([0, 1, 2], [0, 1, 2])) + b
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=1e-5)


def test_norms_match_torch(rng):
    x = torch.as_tensor(rng.normal(size=(2, 5, 6, 4)).astype(np.float32)) * 3 + 1
    g = torch.as_tensor(rng.uniform(0.5, 1.5, 4).astype(np.float32))
    b = torch.as_tensor(rng.uniform(-1, 1, 4).astype(np.float32))
    m = torch.as_tensor(rng.uniform(-1, 1, 4).astype(np.float32))
    v = torch.as_tensor(rng.uniform(0.5, 2, 4).astype(np.float32))
    xc = x.permute(0, 3, 1, 2)
    np.testing.assert_allclose(tf_ops.instance_norm(x, g, b).permute(0, 3, 1, 2).numpy(),
                               F.instance_norm(xc, weight=g, bias=b, eps=1e-3).numpy(), atol=1e-5)
    np.testing.assert_allclose(tf_ops.batch_norm(x, g, b, m, v).permute(0, 3, 1, 2).numpy(),
                               F.batch_norm(xc, m, v, g, b, False, 0.0, 1e-3).numpy(), atol=1e-5)
    np.testing.assert_allclose(tf_ops.batch_norm(x, g, b, m, v, training=True).permute(0, 3, 1, 2).numpy(),
                               F.batch_norm(xc, None, None, g, b, True, 0.0, 1e-3).numpy(), atol=1e-5)


def test_avg_pool_valid_floors_odd_sizes(rng):
    x = torch.as_tensor(rng.normal(size=(3, 7, 9, 1)).astype(np.float32))
    got = tf_ops.avg_pool_2x2_valid(x)
    want = F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert got.shape == (3, 3, 4, 1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-6)


def test_resize_bilinear_is_half_pixel_and_matches_torch(rng):
    x = torch.as_tensor(rng.normal(size=(2, 5, 7, 2)).astype(np.float32))
    got = tf_ops.resize_bilinear(x, 40, 56)
    want = F.interpolate(x.permute(0, 3, 1, 2), size=(40, 56), mode='bilinear',
                         align_corners=False).permute(0, 2, 3, 1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-5)
    np.testing.assert_allclose(oracle.upflow8(x).numpy(), 8 * want.numpy(), atol=1e-4)


def test_upsample_flow_uniform_mask_is_box_filter_of_8x_flow(rng):
    """With all-equal logits the convex combination is the 3x3 zero-padded mean of 8*flow,
    replicated over each 8x8 block (reference model.py:51-66)."""
    flow = torch.as_tensor(rng.normal(size=(1, 4, 5, 2)).astype(np.float32))
    mask = torch.zeros((1, 4, 5, 576))
    up = upsample_flow(flow, mask)
    box = F.avg_pool2d(8 * flow.permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=True).permute(0, 2, 3, 1)
    want = box.repeat_interleave(8, dim=1).repeat_interleave(8, dim=2)
    assert up.shape == (1, 32, 40, 2)
    np.testing.assert_allclose(up.numpy(), want.numpy(), atol=1e-5)


def test_upsample_flow_one_hot_mask_selects_neighbour(rng):
    """mask channel (i*8 + j)*9 + k with k = ky*3 + kx."""
    flow = torch.as_tensor(rng.normal(size=(1, 3, 3, 2)).astype(np.float32))
    mask = torch.full((1, 3, 3, 576), -1e4)
    i, j, k = 5, 2, 7                                               # k=7 -> (ky,kx) = (2,1): pixel below
    mask[..., (i * 8 + j) * 9 + k] = 1e4
    up = upsample_flow(flow, mask)
    np.testing.assert_allclose(up[0, 1 * 8 + i, 1 * 8 + j].numpy(), 8 * flow[0, 2, 1].numpy(), rtol=1e-5)


# ---------------------------------------------------------------- correlation
def test_corr_block_pyramid_shapes_and_values(rng):
    f1 = torch.as_tensor(rng.normal(size=(2, 8, 12, 16)).astype(np.float32))
    f2 = torch.as_tensor(rng.normal(size=(2, 8, 12, 16)).astype(np.float32))
    cb = oracle.CorrBlock(f1, f2, 4, 4)
    assert [tuple(p.shape) for p in cb.corr_pyramid] == [(192, 8, 12, 1), (192, 4, 6, 1), (192, 2, 3, 1),
                                                         (192, 1, 1, 1)]
    q, t = 5 * 12 + 7, (3, 4)
    want = float((f1[1, 5, 7] * f2[1, t[0], t[1]]).sum() / 4.0)
    np.testing.assert_allclose(float(cb.corr_pyramid[0][96 + q, t[0], t[1], 0]), want, rtol=1e-5)


def test_pooling_commutes_with_the_dot_product(rng):
    """Identity the HIP corr_build relies on: level l == <fmap1, avgpool_l(fmap2)> / sqrt(C)."""
    f1 = torch.as_tensor(rng.normal(size=(1, 9, 13, 32)).astype(np.float32)).double()
    f2 = torch.as_tensor(rng.normal(size=(1, 9, 13, 32)).astype(np.float32)).double()
    cb = oracle.CorrBlock(f1, f2, 3, 4)
    p = f2
    for l in range(1, 3):
        p = tf_ops.avg_pool_2x2_valid(p)
        alt = oracle.CorrBlock.correlation(f1, torch.zeros_like(f1))  # shape helper only
        del alt
        want = torch.einsum('bqc,btc->bqt', f1.reshape(1, -1, 32), p.reshape(1, -1, 32)) / np.sqrt(32.0)
        got = cb.corr_pyramid[l].reshape(1, 9 * 13, -1)
        np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-12)


def test_retrieve_iteration0_level0_is_zero_and_axis_quirk(rng):
    f1 = torch.as_tensor(rng.normal(size=(1, 16, 16, 8)).astype(np.float32))
    f2 = torch.as_tensor(rng.normal(size=(1, 16, 16, 8)).astype(np.float32))
    cb = oracle.CorrBlock(f1, f2, 4, 4)
    grid = oracle.coords_grid(1, 16, 16)
    out0 = cb.retrieve(grid)
    assert out0.shape == (1, 16, 16, 324)
    assert torch.all(out0[..., :81] == 0)
    assert torch.any(out0[..., 81:] != 0)
    out = cb.retrieve(grid + 0.25)
    img = cb.corr_pyramid[0][8 * 16 + 8, :, :, 0]
    a, b, r = 6, 1, 4                                                # a offsets x, b offsets y
    sx, sy = 8.25 + (a - r), 8.25 + (b - r)
    x0, y0 = int(sx), int(sy)
    fx, fy = sx - x0, sy - y0
    want = ((1 - fy) * (1 - fx) * img[y0, x0] + (1 - fy) * fx * img[y0, x0 + 1]
            + fy * (1 - fx) * img[y0 + 1, x0] + fy * fx * img[y0 + 1, x0 + 1])
    np.testing.assert_allclose(float(out[0, 8, 8, a * 9 + b]), float(want), rtol=1e-5)


# ---------------------------------------------------------------- layers / model
def test_sep_conv_gru_matches_independent_torch_convs(rng):
    wts = wm.init_weights('raft', seed=2, perturb=True)
    w = W(wts)
    h = torch.as_tensor(np.tanh(rng.normal(size=(1, 6, 7, 128))).astype(np.float32))
    x = torch.as_tensor(rng.normal(size=(1, 6, 7, 256)).astype(np.float32))
    got = sep_conv_gru(w, 'update_block/gru', h, x)

    def conv(name, inp, pad):
        k = torch.as_tensor(wts[f'update_block/gru/{name}/kernel']).permute(3, 2, 0, 1)
        b = torch.as_tensor(wts[f'update_block/gru/{name}/bias'])
        return F.conv2d(inp.permute(0, 3, 1, 2), k, b, padding=pad).permute(0, 2, 3, 1)

    hh = h
    for s, pad in (('1', (0, 2)), ('2', (2, 0))):
        hx = torch.cat([hh, x], -1)
        z = torch.sigmoid(conv(f'convz{s}', hx, pad))
        r = torch.sigmoid(conv(f'convr{s}', hx, pad))
        q = torch.tanh(conv(f'convq{s}', torch.cat([r * hh, x], -1), pad))
        hh = (1 - z) * hh + z * q
    np.testing.assert_allclose(got.numpy(), hh.numpy(), atol=1e-5)


def test_basic_update_block_shapes_and_mask_scale(rng):
    wts = wm.init_weights('raft', seed=0)
    w = W(wts)
    net = torch.zeros((1, 4, 5, 128))
    inp = torch.zeros((1, 4, 5, 128))
    corr = torch.as_tensor(rng.normal(size=(1, 4, 5, 324)).astype(np.float32))
    flow = torch.zeros((1, 4, 5, 2))
    n, m, d = basic_update_block(w, 'update_block', net, inp, corr, flow)
    assert n.shape == (1, 4, 5, 128) and m.shape == (1, 4, 5, 576) and d.shape == (1, 4, 5, 2)


def test_encoder_output_shapes_and_stride8(rng):
    wts = wm.init_weights('raft', seed=0)
    w = W(wts)
    x = torch.as_tensor(rng.uniform(-1, 1, size=(1, 64, 96, 3)).astype(np.float32))
    f1, f2 = encoder(w, 'fnet', [x, x])
    assert f1.shape == (1, 8, 12, 256)
    np.testing.assert_allclose(f1.numpy(), f2.numpy(), atol=1e-6)   # instance norm is per sample
    assert encoder(w, 'cnet', x).shape == (1, 8, 12, 256)
    ws = W(wm.init_weights('small', seed=0))
    assert encoder(ws, 'fnet', x).shape == (1, 8, 12, 128)
    assert encoder(ws, 'cnet', x).shape == (1, 8, 12, 160)


@pytest.mark.parametrize('variant', ['raft', 'small'])
def test_model_output_is_list_of_iters_flows(variant):
    """reference tests/test_model.py:44-77 (shape contract; 64x96 exercises the degenerate 1x1 level)."""
    rng = np.random.default_rng(1)
    i1 = rng.normal(size=(2, 64, 96, 3)).astype(np.float32)
    i2 = rng.normal(size=(2, 64, 96, 3)).astype(np.float32)
    cls = oracle.RAFT if variant == 'raft' else oracle.SmallRAFT
    model = cls(wm.init_weights(variant, seed=0), iters=2, iters_pred=3)
    out = model([i1, i2], training=True)
    assert len(out) == 2 and all(o.shape == (2, 64, 96, 2) for o in out)
    out = model([i1, i2], training=False)
    assert len(out) == 3 and all(o.shape == (2, 64, 96, 2) and np.isfinite(o).all() for o in out)


def test_model_batch_elements_are_independent():
    """Pairs are independent end to end (basis of the data-parallel sharding)."""
    rng = np.random.default_rng(3)
    i1 = rng.uniform(0, 255, (2, 64, 64, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (2, 64, 64, 3)).astype(np.float32)
    model = oracle.RAFT(wm.init_weights('raft', seed=0), iters_pred=2)
    both = model([i1, i2])[-1]
    one = model([i1[1:], i2[1:]])[-1]
    np.testing.assert_allclose(both[1:], one, atol=2e-4)


def test_conditioning_fixture_matches_its_generator():
    """The committed conditioning fixture is reproducible from the committed script (smallest case)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('mk', os.path.join(GOLDEN, 'make_conditioning.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    with open(os.path.join(GOLDEN, 'conditioning.json')) as f:
        cond = json.load(f)
    got = mk.run('small', 64, 96, 12, 0, 'default')
    want = cond['small_64x96_seed0_it12']
    assert len(got['epe32v64']) == 12
    assert max(got['epe32v64']) <= 1e-3 and max(want['epe32v64']) <= 1e-3
    np.testing.assert_allclose(got['max_abs_flow'], want['max_abs_flow'], rtol=1e-3)
    # the conditioned regime (flow head of tf_raft_amd.weights.condition_weights): smallest committed case
    got = mk.run('small', 256, 256, 4, 0, 'conditioned')
    want = cond[mk.case_key('small', 256, 256, 4, 0, 'conditioned')]
    np.testing.assert_allclose(got['max_abs_flow'], want['max_abs_flow'], rtol=1e-3)
    assert max(got['epe32v64']) <= 1e-4
    # every committed north-star case is well conditioned for all of its iterations, and its flow stays inside (0, 8) px
    # at full resolution = (0, 1) feature pixels: no lookup tap can cross an integer
    ns = [k for k in cond if k.endswith('_conditioned')]
    assert len([k for k in ns if k.startswith('raft_448x512')]) >= 3 and len([k for k in ns if k.startswith('small_448x512')]) >= 3
    for k in ns:
        assert max(cond[k]['epe32v64']) <= 2e-4, k
        assert max(cond[k]['max_abs_flow']) < 8.0, k
