"""First slice of the training step (reference model.py:126-144, BASELINE config 5): the backward HIP kernels of
``tf_raft_amd.grad`` against torch autograd through the CPU ORACLE's forward functions (``oracle.sequence_loss`` restated
with torch ops, ``oracle.CorrBlock.retrieve``, ``oracle.tf_ops.conv2d``).  GPU only.

Tolerances: the loss gradient is exact up to one rounding; the lookup's coordinate gradient sums 324 products per pixel
(fp32, different order than autograd): 1e-5 relative to its scale; the pyramid gradient is a sum of at most 4 weight
products, each weight off by at most one ulp of the fp32 coordinate it is derived from; convolution gradients are long fp32 dot products over pixels: compared with
float64 autograd at 2e-5 x sqrt(K) scale (reported)."""
import numpy as np
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu

WINOGRAD_TESTS = ('test_training_convolutions_on_winograd_match_the_direct_kernels', 'test_full_train_step_matches_reference_semantics',
                  'test_train_step_runs_device_resident_with_dropout_and_bf16_tape')


@pytest.fixture(autouse=True)
def _train_conv_algorithm(request):
    """The per-operator tolerances of this file are those of the DIRECT fp32 kernels (a convolution's rounding error is then a
    plain fp32 dot product's): they run with grad.TRAIN_WINOGRAD off.  The tests named in WINOGRAD_TESTS run the product default
    (F(2x2, 3x3) / F(4, 5) forward and input-gradient convolutions) against bounds that include the Winograd noise."""
    from tf_raft_amd import grad
    old = grad.TRAIN_WINOGRAD
    grad.TRAIN_WINOGRAD = request.node.name.split('[')[0] in WINOGRAD_TESTS
    grad.clear_pack_cache()
    yield
    grad.TRAIN_WINOGRAD = old
    grad.clear_pack_cache()


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _torch_sequence_loss(flow_gt, valid, preds, gamma, max_flow):
    """oracle/losses.py sequence_loss (reference losses.py:4-21) with torch ops, so autograd can differentiate it."""
    mag = torch.sqrt((flow_gt ** 2).sum(-1))
    v = (valid & (mag < max_flow)).to(flow_gt.dtype)[..., None]
    n = len(preds)
    loss = 0.0
    for i, p in enumerate(preds):
        loss = loss + gamma ** (n - i - 1) * torch.mean(v * torch.abs(p - flow_gt))
    return loss


@pytest.mark.parametrize('shape,n_pred', [((1, 5, 7), 1), ((2, 64, 96), 12), ((1, 368, 496), 12)])
def test_sequence_loss_grad_matches_autograd(rng, shape, n_pred):
    from oracle import losses as oracle
    from tf_raft_amd import grad
    flow_gt = (rng.normal(size=shape + (2,)) * 4).astype(np.float32)
    flow_gt.reshape(-1, 2)[::53] *= 200.0                      # beyond max_flow: masked
    valid = rng.uniform(size=shape) < 0.8
    preds = [(flow_gt + rng.normal(size=shape + (2,)) * (3.0 / (i + 1))).astype(np.float32) for i in range(n_pred)]
    preds[0][0, 0, 0, 0] = flow_gt[0, 0, 0, 0]                 # |.| at 0: gradient 0 in TF and in torch
    tp = [torch.tensor(p, dtype=torch.float64, requires_grad=True) for p in preds]
    loss = _torch_sequence_loss(torch.tensor(flow_gt, dtype=torch.float64), torch.tensor(valid), tp, 0.8, 400)
    np.testing.assert_allclose(float(loss), oracle.sequence_loss((flow_gt, valid), preds), rtol=1e-5)   # same function
    loss.backward()
    got = grad.sequence_loss_grad((flow_gt, valid), preds, gamma=0.8, max_flow=400)
    assert len(got) == n_pred
    worst = 0.0
    for g, t in zip(got, tp):
        want = t.grad.numpy()
        worst = max(worst, float(np.abs(_np(g) - want).max() / max(np.abs(want).max(), 1e-30)))
    report(f'sequence_loss grad {shape} n={n_pred}', worst_rel=worst)
    assert worst <= 1e-6
    assert _np(got[0])[0, 0, 0, 0] == 0.0
    scaled = grad.sequence_loss_grad((flow_gt, valid), preds, upstream=2.5)
    np.testing.assert_allclose(_np(scaled[-1]), 2.5 * _np(got[-1]), rtol=1e-6)


@pytest.mark.parametrize('radius,shape,sigma', [(4, (2, 8, 12, 32), 2.0), (4, (1, 16, 24, 32), 40.0), (3, (1, 16, 24, 32), 1.0),
                                               (4, (1, 46, 62, 32), 3.0)])
def test_corr_lookup_backward_matches_autograd(rng, radius, shape, sigma):
    """(1, 46, 62) is the feature-map size of the reference's 368 x 496 training crops (configs/train_chairs.yml)."""
    import oracle
    from tf_raft_amd import grad
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w, C = shape
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    dev = CorrBlock(f1, f2, 4, radius)
    ref = oracle.CorrBlock(torch.tensor(f1, dtype=torch.float64), torch.tensor(f2, dtype=torch.float64), 4, radius)
    pyr = [lvl.detach().clone().requires_grad_(True) for lvl in ref.corr_pyramid]
    ref.corr_pyramid = pyr
    for l in range(4):
        dev._set_level(l, pyr[l].detach().to(torch.float32))    # the same volume on both sides
    grid = oracle.coords_grid(B, h, w).numpy()
    coords_np = (grid + rng.normal(scale=sigma, size=grid.shape)).astype(np.float32)
    coords_np[0, 0, 0] = [3.0, 2.5]                              # an exactly integer x: zero weights, zero gradient
    coords = torch.tensor(coords_np, dtype=torch.float64, requires_grad=True)
    out = ref.retrieve(coords)
    d_out = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(torch.tensor(d_out, dtype=torch.float64))
    d_coords, d_pyr = grad.corr_lookup_backward(dev, coords_np, d_out)
    want_c = coords.grad.numpy()
    err_c = float(np.abs(_np(d_coords) - want_c).max())
    scale_c = float(np.abs(want_c).max())
    report(f'lookup backward r={radius} {shape} sigma={sigma} coords', max_abs=err_c, scale=scale_c)
    assert err_c <= 1e-5 * max(1.0, scale_c)
    levels = dev.untile_pyramid(d_pyr)
    for l in range(4):
        want = pyr[l].grad.numpy()
        err = float(np.abs(_np(levels[l]) - want).max())
        report(f'lookup backward r={radius} {shape} level {l} pyramid', max_abs=err, scale=float(np.abs(want).max()),
               nonzero=float((want != 0).mean()))
        # the weights ceil(g) - g, g - floor(g) come from fp32 coordinates: an absolute error of one ulp of the largest
        # coordinate (autograd runs in float64), times the upstream gradient they multiply
        tol = 2.0 * float(np.spacing(np.float32(np.abs(coords_np).max() + radius))) * float(np.abs(d_out).max())
        assert err <= max(tol, 2e-6), (err, tol)
    # accumulation over loop iterations: a second call adds to the same buffer; coords-only mode leaves it alone
    before = d_pyr.clone()
    _, d_pyr2 = grad.corr_lookup_backward(dev, coords_np, d_out, d_pyramid=d_pyr)
    assert d_pyr2 is d_pyr
    np.testing.assert_allclose(_np(d_pyr), 2 * _np(before), rtol=1e-6, atol=1e-7)
    dc2, none = grad.corr_lookup_backward(dev, coords_np, d_out, want_pyramid_grad=False)
    assert none is None
    np.testing.assert_array_equal(_np(dc2), _np(d_coords))      # deterministic


@pytest.mark.parametrize('ksize,cin,cout,shape,relu', [((3, 3), 128, 64, (2, 9, 13), True), ((1, 1), 324 + 28, 256, (1, 8, 16), True),
                                                         ((1, 5), 256, 128, (1, 7, 20), False), ((5, 1), 64, 192, (2, 12, 6), False),
                                                         ((3, 3), 256, 192, (1, 46, 62), True)])
def test_conv2d_backward_matches_autograd(rng, ksize, cin, cout, shape, relu):
    from oracle import tf_ops
    from tf_raft_amd import grad
    kh, kw = ksize
    B, H, W = shape
    x = rng.normal(size=(B, H, W, cin)).astype(np.float32)
    kernel = (rng.normal(size=(kh, kw, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.normal(size=(cout,)).astype(np.float32)
    dy = rng.normal(size=(B, H, W, cout)).astype(np.float32)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tk = torch.tensor(kernel, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(bias, dtype=torch.float64, requires_grad=True)
    y = tf_ops.conv2d(tx, tk, tb)
    if relu:
        y = torch.relu(y)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    dx, dk, db = grad.conv2d_backward(x, kernel, dy, y=y.detach().to(torch.float32).numpy() if relu else None)
    M = B * H * W
    for name, got, want, k_len in (('dx', dx, tx.grad, kh * kw * cout), ('d_kernel', dk, tk.grad, M), ('d_bias', db, tb.grad, M)):
        want = want.numpy()
        err = float(np.abs(_np(got) - want).max())
        tol = 5e-6 * max(1.0, float(np.abs(want).max()))          # fp32 sums of k_len products vs float64 autograd
        report(f'conv backward {ksize} {cin}->{cout} {shape} {name}', max_abs=err, scale=float(np.abs(want).max()), tol=tol)
        assert got.shape == want.shape
        assert err <= tol
    dx2, dk2, db2 = grad.conv2d_backward(x, kernel, dy, y=y.detach().to(torch.float32).numpy() if relu else None)
    np.testing.assert_array_equal(_np(dk2), _np(dk))             # deterministic reductions
    np.testing.assert_array_equal(_np(db2), _np(db))


def test_backward_argument_checks():
    from tf_raft_amd import _dev
    lib = _dev.lib()
    assert lib.raft_conv2d_wgrad_f32(None, 4, 4, None, 4, 4, 1, 4, 4, 3, 3, None, None, None, None) == -1
    assert lib.raft_conv2d_wgrad_workspace_floats(0, 4, 1, 4, 4, 3, 3) == 0
    assert lib.raft_relu_backward_f32(None, None, None, 4, None) == -1
    assert lib.raft_sequence_loss_grad_f32(None, None, None, 8, 1, 4, 0.8, 400.0, 1.0, None, None) == -1


@pytest.mark.parametrize('shape', [(1, 16, 24), (2, 9, 13)])
def test_basic_update_block_backward_matches_autograd(rng, shape):
    """Second slice: one whole BasicUpdateBlock call (reference update.py:128-153: motion encoder, SepConvGRU, flow head,
    mask head) -- forward in training form against the float64 oracle, then the backward of all three outputs against
    torch autograd through ``oracle.layers.basic_update_block``: gradients w.r.t. the four inputs and all 30 kernels and
    biases."""
    from oracle.layers import W, basic_update_block
    from tf_raft_amd import grad
    from tf_raft_amd import weights as wm
    B, h, w = shape
    wts = {k: v for k, v in wm.init_weights('raft', seed=7, perturb=True).items() if k.startswith('update_block')}
    net = np.tanh(rng.normal(size=(B, h, w, 128))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, 128)), 0).astype(np.float32)
    corr = rng.normal(size=(B, h, w, 324)).astype(np.float32)
    flow = (rng.normal(size=(B, h, w, 2)) * 2).astype(np.float32)
    ow = W(wts, torch.float64)
    for t in ow.t.values():
        t.requires_grad_(True)
    tin = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (net, inp, corr, flow)]
    rn, rm, rd = basic_update_block(ow, 'update_block', *tin)
    gn, gm, gd, saved = grad.basic_update_block_forward(wts, net, inp, corr, flow)
    for name, got, want in (('net', gn, rn), ('mask', gm, rm), ('delta', gd, rd)):
        err = float(np.abs(_np(got) - want.detach().numpy()).max())
        report(f'update block training forward {shape} {name}', max_abs_vs_f64=err)
        assert err <= 5e-5
    d_net = rng.normal(size=rn.shape).astype(np.float32)
    d_mask = (rng.normal(size=rm.shape) * 0.1).astype(np.float32)
    d_delta = rng.normal(size=rd.shape).astype(np.float32)
    torch.autograd.backward([rn, rm, rd], [torch.tensor(a, dtype=torch.float64) for a in (d_net, d_mask, d_delta)])
    din, dw = grad.basic_update_block_backward(wts, saved, d_net, d_mask, d_delta)
    worst = 0.0
    for name, t in zip(('net', 'inp', 'corr', 'flow'), tin):
        want = t.grad.numpy()
        rel = float(np.abs(_np(din[name]) - want).max() / max(1.0, np.abs(want).max()))
        report(f'update block backward {shape} d_{name}', rel_err=rel, scale=float(np.abs(want).max()))
        worst = max(worst, rel)
        assert _np(din[name]).shape == want.shape
    assert len(dw) == 30
    for name, got in sorted(dw.items()):
        want = ow.t[name].grad.numpy()
        assert _np(got).shape == want.shape, name
        rel = float(np.abs(_np(got) - want).max() / max(1.0, np.abs(want).max()))
        worst = max(worst, rel)
        if rel > 1e-5:
            report(f'update block backward {shape} {name}', rel_err=rel, scale=float(np.abs(want).max()))
        assert rel <= 5e-5, (name, rel)
    report(f'update block backward {shape}', worst_rel_err_over_34_gradients=worst)
    assert worst <= 5e-5


@pytest.mark.parametrize('shape', [(2, 7, 9), (1, 6, 8)])
def test_upsample_convex_backward_matches_autograd(rng, shape):
    from oracle.model import upsample_flow
    from tf_raft_amd import grad
    B, h, w = shape
    flow = rng.normal(size=(B, h, w, 2)).astype(np.float32)
    mask = rng.normal(size=(B, h, w, 576)).astype(np.float32)
    d_up = rng.normal(size=(B, 8 * h, 8 * w, 2)).astype(np.float32)
    tf, tm = torch.tensor(flow, dtype=torch.float64, requires_grad=True), torch.tensor(mask, dtype=torch.float64, requires_grad=True)
    upsample_flow(tf, tm).backward(torch.tensor(d_up, dtype=torch.float64))
    d_flow, d_mask = grad.upsample_flow_backward(flow, mask, d_up)
    for name, got, want in (('d_flow', d_flow, tf.grad), ('d_mask', d_mask, tm.grad)):
        want = want.numpy()
        rel = float(np.abs(_np(got) - want).max() / max(1.0, np.abs(want).max()))
        report(f'upsample_convex backward {shape} {name}', rel_err=rel, scale=float(np.abs(want).max()))
        assert rel <= 1e-5


def test_loop_backward_through_time_matches_autograd(rng):
    """Third slice: the prediction loop of RAFT.call (reference model.py:91-109) in training form, 3 iterations, and its
    backward through time for the gradient of ``sequence_loss``: lookup -> update block -> coords update -> convex
    upsampling, with the gradient flowing through the lookup coordinates of every iteration (the reference has no
    stop_gradient).  Checked against torch autograd through the oracle's loop (float64) on the same correlation volume:
    d loss / d {net0, inp, every level of the volume, all 30 update-block kernels and biases}.  Conditioned weights keep the
    lookup away from its discontinuities."""
    import oracle
    from oracle.layers import W, basic_update_block
    from oracle.model import upsample_flow
    from tf_raft_amd import grad
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w, C, iters = 1, 16, 24, 32, 3
    wts_all = wm.condition_weights('raft', wm.init_weights('raft', seed=3))
    wts = {k: v for k, v in wts_all.items() if k.startswith('update_block')}
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    net0 = np.tanh(rng.normal(size=(B, h, w, 128))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, 128)), 0).astype(np.float32)
    flow_gt = (rng.normal(size=(B, 8 * h, 8 * w, 2)) * 2).astype(np.float32)
    valid = rng.uniform(size=(B, 8 * h, 8 * w)) < 0.9
    # ---- oracle side (float64, autograd)
    ref = oracle.CorrBlock(torch.tensor(f1, dtype=torch.float64), torch.tensor(f2, dtype=torch.float64), 4, 4)
    pyr = [lvl.detach().clone().requires_grad_(True) for lvl in ref.corr_pyramid]
    ref.corr_pyramid = pyr
    ow = W(wts, torch.float64)
    for t in ow.t.values():
        t.requires_grad_(True)
    tnet = torch.tensor(net0, dtype=torch.float64, requires_grad=True)
    tinp = torch.tensor(inp, dtype=torch.float64, requires_grad=True)
    coords0 = oracle.coords_grid(B, h, w, torch.float64)
    coords1, net, preds = coords0.clone(), tnet, []
    for _ in range(iters):
        corr = ref.retrieve(coords1)
        net, mask, delta = basic_update_block(ow, 'update_block', net, tinp, corr, coords1 - coords0)
        coords1 = coords1 + delta
        preds.append(upsample_flow(coords1 - coords0, mask))
    loss = _torch_sequence_loss(torch.tensor(flow_gt, dtype=torch.float64), torch.tensor(valid), preds, 0.8, 400)
    loss.backward()
    # ---- device side
    dev = CorrBlock(f1, f2, 4, 4)
    for l in range(4):
        dev._set_level(l, pyr[l].detach().to(torch.float32))
    gpreds, tape = grad.loop_forward(wts, dev, net0, inp, iters)
    for i in range(iters):
        err = float(np.abs(_np(gpreds[i]) - preds[i].detach().numpy()).max())
        report(f'loop training forward prediction {i}', max_abs_vs_f64=err)
        assert err <= 1e-4
    d_preds = grad.sequence_loss_grad((flow_gt, valid), gpreds, gamma=0.8, max_flow=400)
    d_net0, d_inp, d_pyr, wg = grad.loop_backward(wts, dev, tape, d_preds)
    worst = 0.0

    def cmp(name, got, want):
        nonlocal worst
        want = want.numpy()
        rel = float(np.abs(_np(got) - want).max() / max(np.abs(want).max(), 1e-12))
        report(f'loop backward {name}', rel_err=rel, scale=float(np.abs(want).max()))
        worst = max(worst, rel)
        assert rel <= 2e-3, (name, rel)

    cmp('d_net0', d_net0, tnet.grad)
    cmp('d_inp', d_inp, tinp.grad)
    for l, lvl in enumerate(dev.untile_pyramid(d_pyr)):
        cmp(f'd_pyramid[{l}]', lvl, pyr[l].grad)
    assert len(wg) == 30
    worst_w = 0.0
    for name in sorted(wg):
        # a weight gradient is a signed sum over all pixels and iterations, accumulated in fp32 here and in float64 by
        # autograd: compared in the L2 norm of the tensor.  The flow branch (convf1 / convf2) sits at ~1e-3: with the
        # conditioned weights its input (the flow, ~0.02 px) puts many relu pre-activations within rounding of zero, and a
        # kink decided differently in fp32 and float64 moves that unit's whole contribution; every other layer is < 2e-4
        want = ow.t[name].grad.numpy()
        diff = _np(wg[name]).astype(np.float64) - want
        rel2 = float(np.linalg.norm(diff) / max(np.linalg.norm(want), 1e-30))
        if rel2 > 2e-4:
            report(f'loop backward {name}', rel_l2=rel2, scale=float(np.abs(want).max()))
        worst_w = max(worst_w, rel2)
        assert rel2 <= 5e-3, (name, rel2)
    report('loop backward through time, 3 iterations', worst_rel_err_inputs=worst, worst_rel_l2_weights=worst_w)
    # the weight gradients evaluated per iteration and accumulated (what a bf16 tape does) are the same numbers up to the
    # order of the fp32 pixel sums
    _, _, _, wg_it = grad.loop_backward(wts, dev, tape, d_preds, defer_wgrad=False)
    assert sorted(wg_it) == sorted(wg)
    worst_d = 0.0
    for name in sorted(wg):
        a, b_ = _np(wg[name]).astype(np.float64), _np(wg_it[name]).astype(np.float64)
        worst_d = max(worst_d, float(np.linalg.norm(a - b_) / max(np.linalg.norm(b_), 1e-30)))
    report('deferred vs per-iteration weight gradients', worst_rel_l2=worst_d)
    assert worst_d <= 2e-5


def test_training_convolutions_on_winograd_match_the_direct_kernels(rng):
    """grad.TRAIN_WINOGRAD (default on): one BasicUpdateBlock call forward + backward with the 3x3 / 1x5 / 5x1 layers on the
    F(2x2, 3x3) / F(4, 5) kernels against the same call on the direct kernels -- outputs, input gradients and all 30 weight
    gradients within 2e-5 of the tensor's scale (the direct path itself is within 1.2e-6 of float64 autograd), with NumPy and with
    device-resident parameters (the transforms G g G^T / G' g are then evaluated on the device)."""
    from tf_raft_amd import grad, _dev
    from tf_raft_amd import weights as wm
    B, h, w = 2, 14, 22
    wts = {k: v for k, v in wm.condition_weights('raft', wm.init_weights('raft', seed=3)).items() if k.startswith('update_block')}
    net = np.tanh(rng.normal(size=(B, h, w, 128))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, 128)), 0).astype(np.float32)
    corr = rng.normal(size=(B, h, w, 324)).astype(np.float32)
    flow = rng.normal(size=(B, h, w, 2)).astype(np.float32)
    d_net = rng.normal(size=(B, h, w, 128)).astype(np.float32)
    d_mask = rng.normal(size=(B, h, w, 576)).astype(np.float32)
    d_delta = rng.normal(size=(B, h, w, 2)).astype(np.float32)

    def run(params):
        grad.clear_pack_cache()
        n, m, d, saved = grad.basic_update_block_forward(params, net, inp, corr, flow)
        din, dw = grad.basic_update_block_backward(params, saved, d_net, d_mask, d_delta)
        return [_np(n), _np(m), _np(d)], {k: _np(v) for k, v in din.items()}, {k: _np(v) for k, v in dw.items()}

    assert grad.TRAIN_WINOGRAD
    got = run(wts)
    got_dev = run({k: _dev.to_device(v) for k, v in wts.items()})
    grad.TRAIN_WINOGRAD = False
    want = run(wts)
    worst = 0.0
    for label, g in (('host parameters', got), ('device parameters', got_dev)):
        for a, b_ in zip(g[0], want[0]):
            worst = max(worst, float(np.abs(a - b_).max() / np.abs(b_).max()))
        for part in (1, 2):
            assert sorted(g[part]) == sorted(want[part])
            for k in want[part]:
                worst = max(worst, float(np.abs(g[part][k] - want[part][k]).max() / max(np.abs(want[part][k]).max(), 1e-30)))
        report(f'training convolutions on Winograd vs direct ({label})', worst_rel=worst)
    assert worst <= 5e-5


def test_conv2d_wgrad_multi_is_the_sum_of_the_single_gradients(rng):
    """raft_conv2d_wgrad_multi_f32: the kernel / bias gradient over several (x, dy) pairs in one pixel reduction, against the
    float64 sum of the per-pair gradients, for every kernel shape of the library and ragged tiles."""
    import ctypes as C
    from tf_raft_amd import _dev
    from tf_raft_amd._ffi import check
    lib = _dev.lib()
    dev = _dev.require_gpu()
    for kh, kw, cin, cout, B, H, W, nseg in ((3, 3, 64, 96, 2, 13, 21, 3), (1, 1, 128, 68, 1, 9, 33, 5), (1, 5, 96, 64, 2, 8, 16, 2),
                                             (5, 1, 64, 128, 1, 16, 16, 12), (3, 3, 128, 256, 1, 12, 31, 1)):
        xs = [torch.as_tensor(rng.normal(size=(B, H, W, cin)).astype(np.float32)).to(dev) for _ in range(nseg)]
        dys = [torch.as_tensor(rng.normal(size=(B, H, W, cout)).astype(np.float32)).to(dev) for _ in range(nseg)]
        want_k = np.zeros((kh, kw, cin, cout))
        want_b = np.zeros((cout,))
        for x, dy in zip(xs, dys):
            xd = torch.nn.functional.pad(x.double().cpu(), (0, 0, (kw - 1) // 2, (kw - 1) // 2, (kh - 1) // 2, (kh - 1) // 2))
            dd = dy.double().cpu()
            for ky in range(kh):
                for kx in range(kw):
                    want_k[ky, kx] += torch.einsum('bhwi,bhwo->io', xd[:, ky:ky + H, kx:kx + W], dd).numpy()
            want_b += dd.sum(dim=(0, 1, 2)).numpy()
        ws = torch.empty((int(lib.raft_conv2d_wgrad_workspace_floats(cin, cout, B * nseg, H, W, kh, kw)),), device=dev)
        dk = torch.empty((kh, kw, cin, cout), device=dev)
        db = torch.empty((cout,), device=dev)
        px = (C.c_void_p * nseg)(*[_dev.ptr(t) for t in xs])
        pd = (C.c_void_p * nseg)(*[_dev.ptr(t) for t in dys])
        check(lib.raft_conv2d_wgrad_multi_f32(px, pd, nseg, cin, cin, cout, cout, B, H, W, kh, kw, _dev.ptr(dk), _dev.ptr(db),
                                              _dev.ptr(ws), _dev.stream_ptr()), 'wgrad_multi')
        ek = float(np.abs(_np(dk) - want_k).max() / np.abs(want_k).max())
        eb = float(np.abs(_np(db) - want_b).max() / np.abs(want_b).max())
        report(f'wgrad multi {kh}x{kw} {cin}->{cout} x{nseg}', rel_kernel=ek, rel_bias=eb)
        assert ek <= 2e-5 and eb <= 2e-5
    # argument checks: more segments than the kernel arguments hold
    assert lib.raft_conv2d_wgrad_multi_f32(px, pd, 33, cin, cin, cout, cout, B, H, W, kh, kw, _dev.ptr(dk), _dev.ptr(db),
                                           _dev.ptr(ws), _dev.stream_ptr()) < 0


def _reference_train_steps(wts, batches, iters, lr_fn, wd, clip_norm, n_steps):
    """The reference's train_step (model.py:126-144) restricted to the update block, on the oracle in float64: forward with
    frozen encoders in inference mode, sequence_loss, autograd, tf.clip_by_global_norm, tfa AdamW (tfa 0.11.1: decoupled
    decay not scaled by the learning rate; Keras Adam, epsilon 1e-7).  Returns (losses, updated update-block weights)."""
    import oracle
    from oracle.layers import W, basic_update_block, encoder
    from oracle.model import upsample_flow
    names = sorted(k for k in wts if k.startswith('update_block'))
    var = {k: torch.tensor(wts[k], dtype=torch.float64) for k in names}
    m = {k: torch.zeros_like(v) for k, v in var.items()}
    v2 = {k: torch.zeros_like(v) for k, v in var.items()}
    frozen = W({k: val for k, val in wts.items() if not k.startswith('update_block')}, torch.float64)
    losses = []
    for step in range(n_steps):
        i1, i2, flow_gt, valid = batches[step]
        with torch.no_grad():
            x1 = 2 * (torch.tensor(i1, dtype=torch.float64) / 255.0) - 1.0
            x2 = 2 * (torch.tensor(i2, dtype=torch.float64) / 255.0) - 1.0
            f1, f2 = encoder(frozen, 'fnet', [x1, x2])
            corr_blk = oracle.CorrBlock(f1, f2, 4, 4)
            cnet = encoder(frozen, 'cnet', x1)
            net, inp = torch.tanh(cnet[..., :128]), torch.relu(cnet[..., 128:])
        ow = W({}, torch.float64)
        ow.t = {k: val.clone().requires_grad_(True) for k, val in var.items()}
        B, h, w, _ = net.shape
        coords0 = oracle.coords_grid(B, h, w, torch.float64)
        coords1, preds = coords0.clone(), []
        for _ in range(iters):
            corr = corr_blk.retrieve(coords1)
            net, mask, delta = basic_update_block(ow, 'update_block', net, inp, corr, coords1 - coords0)
            coords1 = coords1 + delta
            preds.append(upsample_flow(coords1 - coords0, mask))
        loss = _torch_sequence_loss(torch.tensor(flow_gt, dtype=torch.float64), torch.tensor(valid), preds, 0.8, 400)
        loss.backward()
        losses.append(float(loss))
        g = {k: ow.t[k].grad for k in names}
        gnorm = torch.sqrt(sum((gg ** 2).sum() for gg in g.values()))
        scale = clip_norm / max(float(gnorm), clip_norm)
        t = step + 1
        lr = lr_fn(step)
        lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        for k in names:
            gk = g[k] * scale
            w_ = var[k] - wd * var[k]
            m[k] = 0.9 * m[k] + 0.1 * gk
            v2[k] = 0.999 * v2[k] + 0.001 * gk * gk
            var[k] = w_ - lr_t * m[k] / (torch.sqrt(v2[k]) + 1e-7)
    return losses, {k: v.numpy() for k, v in var.items()}


def test_train_step_update_block_matches_reference_semantics(rng):
    """RAFT.train_step with trainable='update_block' (reference model.py:126-144, train_sintel.py:83-102 optimizer setup):
    two steps at (1, 64, 96), iters = 3, cyclical learning rate + AdamW + global-norm clipping, against the same procedure
    on the float64 oracle with autograd.  Losses agree to 1e-4 relative; the updated weights are compared through the
    UPDATE they received (Adam's first steps are lr * sign(g)-like, so an update is O(lr) whatever the gradient scale:
    elements whose gradient is at rounding level may take a different sign)."""
    import tf_raft_amd
    from tf_raft_amd import losses, training
    from tf_raft_amd import weights as wm
    B, H, W, iters, n_steps = 1, 64, 96, 3, 2
    wts = wm.condition_weights('raft', wm.init_weights('raft', seed=5))
    batches = []
    for _ in range(n_steps):
        i1 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
        i2 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
        batches.append((i1, i2, (rng.normal(size=(B, H, W, 2)) * 2).astype(np.float32), rng.uniform(size=(B, H, W)) < 0.9))
    lr0, wd, clip = 4e-4, 1e-4, 1.0
    sched = training.CyclicalLearningRate(lr0, 2 * lr0, step_size=1000, scale_fn=training.first_cycle_scaler, scale_mode='cycle')
    assert sched(0) == lr0 and abs(sched(1000) - 2 * lr0) < 1e-12 and sched(2000) == lr0 and sched(2500) == lr0   # tfa triangular, first cycle only
    model = tf_raft_amd.RAFT(weights=wts, iters=iters, iters_pred=4)
    with pytest.raises(ValueError):
        model.compile(optimizer=training.AdamW(wd, sched), clip_norm=clip, trainable='encoders')
    model.compile(optimizer=training.AdamW(wd, sched), clip_norm=clip, loss=losses.sequence_loss, epe=losses.end_point_error,
                  trainable='update_block')
    got_losses = []
    for step in range(n_steps):
        res = model.train_step(batches[step])
        assert set(res) == {'loss', 'epe', 'u1', 'u3', 'u5'}
        got_losses.append(float(model.flow_metrics['loss'].total))
    got_losses = [got_losses[0]] + [b - a for a, b in zip(got_losses, got_losses[1:])]
    want_losses, want_w = _reference_train_steps(wts, batches, iters, sched, wd, clip, n_steps)
    report('train_step losses', got0=got_losses[0], want0=want_losses[0], got1=got_losses[1], want1=want_losses[1])
    np.testing.assert_allclose(got_losses, want_losses, rtol=1e-4)
    new_w = model.get_weights_dict()
    total, bad, num, den = 0, 0, 0.0, 0.0
    for k, ref in want_w.items():
        upd_ref = ref - wts[k].astype(np.float64)
        upd_got = new_w[k].astype(np.float64) - wts[k].astype(np.float64)
        total += upd_ref.size
        bad += int((np.abs(upd_got - upd_ref) > 0.25 * lr0).sum())
        num += float(((upd_got - upd_ref) ** 2).sum())
        den += float((upd_ref ** 2).sum())
        assert np.abs(upd_ref).max() > 0.1 * lr0                      # the reference did move this tensor
    for k, v in wts.items():
        if not k.startswith('update_block'):
            np.testing.assert_array_equal(new_w[k], v)                # frozen encoders
    report('train_step updated weights', frac_elements_off_by_quarter_lr=bad / total, rel_l2_of_update=float(np.sqrt(num / den)))
    assert bad / total <= 0.02
    assert np.sqrt(num / den) <= 0.1
    # the updated weights are live in the inference kernels
    out = model([batches[0][0], batches[0][1]])
    assert len(out) == 4 and np.isfinite(out[-1].numpy()).all()


@pytest.mark.parametrize('shape', [(2, 8, 12, 64), (1, 9, 13, 32), (1, 46, 62, 256)])
def test_corr_build_backward_matches_autograd(rng, shape):
    """Backward of the volume build (reference corr.py:100-114, 154-162: matmul / sqrt(C), three average poolings) for a
    random upstream gradient on every level of the pyramid.  (1, 46, 62, 256) is the training crop's feature map."""
    import oracle
    from tf_raft_amd import grad
    from tf_raft_amd.layers.corr import CorrBlock, tile_maps
    B, h, w, C = shape
    levels = 4 if min(h, w) >= 8 else 3
    f1 = rng.normal(size=shape).astype(np.float32)
    f2 = rng.normal(size=shape).astype(np.float32)
    t1 = torch.tensor(f1, dtype=torch.float64, requires_grad=True)
    t2 = torch.tensor(f2, dtype=torch.float64, requires_grad=True)
    ref = oracle.CorrBlock(t1, t2, levels, 4)
    dev = CorrBlock(f1, f2, levels, 4)
    ups = [rng.normal(size=tuple(lvl.shape)).astype(np.float32) for lvl in ref.corr_pyramid]
    torch.autograd.backward(ref.corr_pyramid, [torch.tensor(u, dtype=torch.float64) for u in ups])
    d_pyr = torch.zeros_like(dev._pyr)
    for l, u in enumerate(ups):                      # the upstream gradient in the library's tiled layout
        d_pyr[dev._off[l]:dev._off[l + 1]] = tile_maps(torch.as_tensor(u[..., 0]).to(d_pyr.device)).reshape(-1)
    d1, d2 = grad.corr_build_backward(dev, d_pyr)
    for name, got, want in (('d_fmap1', d1, t1.grad), ('d_fmap2', d2, t2.grad)):
        want = want.numpy()
        rel = float(np.abs(_np(got) - want).max() / max(1.0, np.abs(want).max()))
        report(f'corr_build backward {shape} {name}', rel_err=rel, scale=float(np.abs(want).max()))
        assert rel <= 5e-6


def test_gemm_and_state_backward(rng):
    from tf_raft_amd import _dev, grad
    from tf_raft_amd._ffi import check
    # strided batched GEMM: C = 0.5 * A^T B + 2 C0 with ragged sizes
    Bt, M, N, K = 2, 70, 45, 37
    a = rng.normal(size=(Bt, K, M)).astype(np.float32)            # A(b, m, k) = a[b, k, m]
    b = rng.normal(size=(Bt, K, N)).astype(np.float32)
    c0 = rng.normal(size=(Bt, M, N)).astype(np.float32)
    a_d, b_d, c_d = _dev.to_device(a), _dev.to_device(b), _dev.to_device(c0.copy())
    check(_dev.lib().raft_gemm_f32(_dev.ptr(a_d), K * M, 1, M, _dev.ptr(b_d), K * N, N, 1, _dev.ptr(c_d), M * N, N, Bt, M, N, K, 0.5, 2.0,
                                   _dev.stream_ptr()), 'gemm')
    want = 0.5 * np.einsum('bkm,bkn->bmn', a.astype(np.float64), b.astype(np.float64)) + 2.0 * c0
    np.testing.assert_allclose(_np(c_d), want, atol=2e-5)
    net0 = np.tanh(rng.normal(size=(1, 5, 7, 128))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(1, 5, 7, 128)), 0).astype(np.float32)
    dn, di = rng.normal(size=net0.shape).astype(np.float32), rng.normal(size=inp.shape).astype(np.float32)
    got = _np(grad.prepare_state_backward(net0, inp, dn, di))
    np.testing.assert_allclose(got[..., :128], dn * (1 - net0 ** 2), rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(got[..., 128:], np.where(inp > 0, di, 0))


@pytest.mark.parametrize('prefix,variant', [('fnet', 'raft'), ('cnet', 'raft'), ('cnet', 'small')])
def test_encoder_backward_matches_autograd(rng, prefix, variant):
    """Fifth slice: the encoders in training form (reference extractor.py:6-49, 88-175): fnet = instance norm over the two
    frames batched together, RAFT cnet = batch norm with BATCH statistics (training=True), SmallRAFT cnet = no norm.
    Forward against the float64 oracle, then the gradient of a random upstream w.r.t. every kernel, bias, gamma, beta
    against autograd."""
    from oracle.layers import W, encoder
    from tf_raft_amd import grad
    from tf_raft_amd import weights as wm
    B, H, Wd = 2, 64, 96
    wts = {k: v for k, v in wm.init_weights(variant, seed=9, perturb=True).items() if k.startswith(prefix + '/')}
    x = (2 * rng.uniform(0, 1, (B, H, Wd, 3)) - 1).astype(np.float32)
    ow = W(wts, torch.float64)
    for k, t in ow.t.items():
        if 'moving' not in k:
            t.requires_grad_(True)
    ref = encoder(ow, prefix, torch.tensor(x, dtype=torch.float64), training=True)
    got, tape = grad.encoder_forward(wts, prefix, x, training=True)
    err = float(np.abs(_np(got) - ref.detach().numpy()).max())
    report(f'encoder training forward {variant} {prefix}', max_abs_vs_f64=err, scale=float(ref.abs().max()))
    assert err <= 1e-4 * max(1.0, float(ref.abs().max()))
    d_out = rng.normal(size=tuple(ref.shape)).astype(np.float32)
    ref.backward(torch.tensor(d_out, dtype=torch.float64))
    g, stats = grad.encoder_backward(wts, prefix, tape, d_out)
    names = sorted(k for k in wts if 'moving' not in k)
    assert sorted(g) == names
    worst = 0.0
    for k in names:
        want = ow.t[k].grad.numpy()
        diff = _np(g[k]).astype(np.float64) - want
        assert _np(g[k]).shape == want.shape, k
        if np.linalg.norm(want) < 1e-9 * np.sqrt(want.size):
            # a bias in front of a normalisation layer: the norm removes it, its true gradient is exactly zero (autograd
            # returns ~1e-17); the fp32 sums leave rounding noise
            assert np.abs(diff).max() <= 5e-3, (k, float(np.abs(diff).max()))     # ~1e-7 of the sum of |terms|
            continue
        rel2 = float(np.linalg.norm(diff) / np.linalg.norm(want))
        worst = max(worst, rel2)
        assert rel2 <= 2e-3, (k, rel2)
    report(f'encoder backward {variant} {prefix}', worst_rel_l2_over_all_parameters=worst, n_parameters=len(names),
           batch_norm_layers=len(stats))
    assert (len(stats) > 0) == (variant == 'raft' and prefix == 'cnet')


def test_full_train_step_matches_reference_semantics(rng):
    """RAFT.train_step as the reference defines it (model.py:126-144): ALL weights trainable, forward in training mode
    (cnet's batch norm on batch statistics), sequence_loss, clip_by_global_norm, AdamW -- one step at (2, 64, 96), iters = 2,
    against the same procedure on the float64 oracle (`oracle.RAFT(...)(training=True)` under autograd).  Compared: the loss,
    the UPDATE of every trainable tensor (O(lr) each: Adam's first step is lr * sign(g)-like), and that the moving statistics
    moved toward the batch statistics."""
    import oracle
    import tf_raft_amd
    from tf_raft_amd import losses, training
    from tf_raft_amd import weights as wm
    B, H, W, iters = 2, 64, 96, 2
    wts = wm.condition_weights('raft', wm.init_weights('raft', seed=6, perturb=True))
    i1 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
    flow_gt = (rng.normal(size=(B, H, W, 2)) * 2).astype(np.float32)
    valid = rng.uniform(size=(B, H, W)) < 0.9
    lr, wd, clip = 4e-4, 1e-4, 1.0
    # ---- reference: oracle in float64 under autograd
    om = oracle.RAFT(wts, iters=iters, dtype=torch.float64)
    names = sorted(k for k in wts if 'moving' not in k)
    for k in names:
        om.w.t[k].requires_grad_(True)
    preds = om([i1, i2], training=True, return_numpy=False)
    assert len(preds) == iters
    loss = _torch_sequence_loss(torch.tensor(flow_gt, dtype=torch.float64), torch.tensor(valid), preds, 0.8, 400)
    loss.backward()
    g = {k: om.w.t[k].grad for k in names}
    gnorm = float(torch.sqrt(sum((v ** 2).sum() for v in g.values())))
    scale = clip / max(gnorm, clip)
    lr_t = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = {}
    for k in names:
        gk = g[k] * scale
        v0 = torch.tensor(wts[k], dtype=torch.float64)
        want[k] = ((v0 - wd * v0) - lr_t * (0.1 * gk) / (torch.sqrt(0.001 * gk * gk) + 1e-7)).numpy()
    # ---- device
    model = tf_raft_amd.RAFT(weights=wts, iters=iters, iters_pred=3)
    model.compile(optimizer=training.AdamW(wd, lr), clip_norm=clip, loss=losses.sequence_loss, epe=losses.end_point_error)
    res = model.train_step((i1, i2, flow_gt, valid))
    report('full train_step', loss=float(res['loss']), want_loss=float(loss), global_norm=gnorm)
    np.testing.assert_allclose(float(res['loss']), float(loss), rtol=1e-4)
    new_w = model.get_weights_dict()
    total, bad, num, den = 0, 0, 0.0, 0.0
    worst_group = {}
    for k in names:
        upd_ref = want[k] - wts[k].astype(np.float64)
        upd_got = new_w[k].astype(np.float64) - wts[k].astype(np.float64)
        total += upd_ref.size
        nb = int((np.abs(upd_got - upd_ref) > 0.25 * lr).sum())
        bad += nb
        num += float(((upd_got - upd_ref) ** 2).sum())
        den += float((upd_ref ** 2).sum())
        grp = k.split('/')[0]
        worst_group[grp] = max(worst_group.get(grp, 0.0), nb / upd_ref.size)
    report('full train_step updated weights', frac_elements_off_by_quarter_lr=bad / total, rel_l2_of_update=float(np.sqrt(num / den)),
           **{f'worst_tensor_frac_{k}': v for k, v in worst_group.items()})
    # parameters whose true gradient is exactly zero (conv biases in front of a norm) get a sign from rounding noise on
    # either side: they are the only elements allowed to differ, and they are < 0.1 % of all elements
    assert bad / total <= 2e-3
    assert np.sqrt(num / den) <= 0.1
    moved = [k for k in wts if k.endswith('moving_mean') and not np.array_equal(new_w[k], wts[k])]
    assert len(moved) == 15                                               # every batch-norm layer of cnet
    out = model([i1, i2])
    assert len(out) == 3 and np.isfinite(out[-1].numpy()).all()


def test_small_update_block_backward_matches_autograd(rng):
    """SmallUpdateBlock (reference update.py:109-125: SmallMotionEncoder, 3x3 ConvGRU, FlowHead(128), no mask) in training
    form and its backward against autograd."""
    from oracle.layers import W, small_update_block
    from tf_raft_amd import grad
    from tf_raft_amd import weights as wm
    B, h, w = 2, 9, 13
    wts = {k: v for k, v in wm.init_weights('small', seed=8, perturb=True).items() if k.startswith('update_block')}
    net = np.tanh(rng.normal(size=(B, h, w, 96))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, 64)), 0).astype(np.float32)
    corr = rng.normal(size=(B, h, w, 196)).astype(np.float32)
    flow = (rng.normal(size=(B, h, w, 2)) * 2).astype(np.float32)
    ow = W(wts, torch.float64)
    for t in ow.t.values():
        t.requires_grad_(True)
    tin = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (net, inp, corr, flow)]
    rn, rm, rd = small_update_block(ow, 'update_block', *tin)
    gn, gm, gd, saved = grad.update_block_forward(wts, net, inp, corr, flow, variant='small')
    assert rm is None and gm is None
    assert float(np.abs(_np(gn) - rn.detach().numpy()).max()) <= 5e-5 and float(np.abs(_np(gd) - rd.detach().numpy()).max()) <= 5e-5
    d_net, d_delta = rng.normal(size=rn.shape).astype(np.float32), rng.normal(size=rd.shape).astype(np.float32)
    torch.autograd.backward([rn, rd], [torch.tensor(a, dtype=torch.float64) for a in (d_net, d_delta)])
    din, dw = grad.update_block_backward(wts, saved, d_net, None, d_delta)
    worst = 0.0
    for name, t in zip(('net', 'inp', 'corr', 'flow'), tin):
        want = t.grad.numpy()
        worst = max(worst, float(np.abs(_np(din[name]) - want).max() / max(1.0, np.abs(want).max())))
    assert len(dw) == 18
    for name, got in dw.items():
        want = ow.t[name].grad.numpy()
        assert _np(got).shape == want.shape, name
        worst = max(worst, float(np.abs(_np(got) - want).max() / max(1.0, np.abs(want).max())))
    report('small update block backward', worst_rel_err_over_22_gradients=worst)
    assert worst <= 5e-5


def test_upflow8_backward_matches_autograd(rng):
    from oracle.corr import upflow8
    from tf_raft_amd import grad
    for B, h, w in ((2, 5, 7), (1, 1, 3)):
        flow = torch.tensor(rng.normal(size=(B, h, w, 2)), dtype=torch.float64, requires_grad=True)
        d_up = rng.normal(size=(B, 8 * h, 8 * w, 2)).astype(np.float32)
        upflow8(flow).backward(torch.tensor(d_up, dtype=torch.float64))
        got = _np(grad.upflow8_backward(d_up, B, h, w))
        rel = float(np.abs(got - flow.grad.numpy()).max() / np.abs(flow.grad.numpy()).max())
        report(f'upflow8 backward {(B, h, w)}', rel_err=rel)
        assert rel <= 1e-5


def test_small_raft_train_step_matches_reference_semantics(rng):
    """SmallRAFT.train_step (reference model.py:173-226 forward, 126-144 step): one step at (2, 64, 96), iters = 2, all weights
    trainable, against the float64 oracle under autograd."""
    import oracle
    import tf_raft_amd
    from tf_raft_amd import losses, training
    from tf_raft_amd import weights as wm
    B, H, W, iters = 2, 64, 96, 2
    wts = wm.condition_weights('small', wm.init_weights('small', seed=4, perturb=True))
    i1 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
    flow_gt = (rng.normal(size=(B, H, W, 2)) * 2).astype(np.float32)
    valid = rng.uniform(size=(B, H, W)) < 0.9
    lr, wd, clip = 4e-4, 1e-4, 1.0
    om = oracle.SmallRAFT(wts, iters=iters, dtype=torch.float64)
    names = sorted(wts)
    for k in names:
        om.w.t[k].requires_grad_(True)
    preds = om([i1, i2], training=True, return_numpy=False)
    loss = _torch_sequence_loss(torch.tensor(flow_gt, dtype=torch.float64), torch.tensor(valid), preds, 0.8, 400)
    loss.backward()
    g = {k: om.w.t[k].grad for k in names}
    gnorm = float(torch.sqrt(sum((v ** 2).sum() for v in g.values())))
    scale = clip / max(gnorm, clip)
    lr_t = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
    model = tf_raft_amd.SmallRAFT(weights=wts, iters=iters, iters_pred=3)
    model.compile(optimizer=training.AdamW(wd, lr), clip_norm=clip, loss=losses.sequence_loss, epe=losses.end_point_error)
    res = model.train_step((i1, i2, flow_gt, valid))
    np.testing.assert_allclose(float(res['loss']), float(loss), rtol=1e-4)
    new_w = model.get_weights_dict()
    total, bad, num, den = 0, 0, 0.0, 0.0
    for k in names:
        gk = g[k] * scale
        v0 = torch.tensor(wts[k], dtype=torch.float64)
        want = ((v0 - wd * v0) - lr_t * (0.1 * gk) / (torch.sqrt(0.001 * gk * gk) + 1e-7)).numpy()
        upd_ref, upd_got = want - wts[k], new_w[k].astype(np.float64) - wts[k]
        total += upd_ref.size
        bad += int((np.abs(upd_got - upd_ref) > 0.25 * lr).sum())
        num += float(((upd_got - upd_ref) ** 2).sum())
        den += float((upd_ref ** 2).sum())
    report('SmallRAFT train_step', loss=float(res['loss']), frac_elements_off_by_quarter_lr=bad / total,
           rel_l2_of_update=float(np.sqrt(num / den)))
    assert bad / total <= 2e-3 and np.sqrt(num / den) <= 0.1
    assert np.isfinite(model([i1, i2])[-1].numpy()).all()


# ------------------------------------------------------------------------------------------------------------------------
# round 3: device-resident parameters, bf16 tape storage, dropout
# ------------------------------------------------------------------------------------------------------------------------
def test_loop_with_device_parameters_and_bf16_tape(rng):
    """(a) grad.* with DEVICE tensors as parameters (what train_step keeps: no host packing) gives bit-for-bit the gradients of
    the NumPy-parameter path; (b) ``tape_dtype='bf16'`` (BASELINE configs[4]: bf16 storage, fp32 arithmetic) stores every
    large activation of the tape as bf16: the forward predictions are unchanged (the forward computes in fp32 and only the
    saved copies are narrowed) and the gradients stay within the stated tolerances of the fp32-tape gradients --
    5e-3 relative (max norm) on d_net0 / d_inp, 2e-2 in the L2 norm on every weight gradient."""
    from tf_raft_amd import _dev, grad
    from tf_raft_amd import weights as wm
    from tf_raft_amd.layers.corr import CorrBlock
    B, h, w, C, iters = 2, 16, 24, 32, 3
    wts_all = wm.condition_weights('raft', wm.init_weights('raft', seed=3, perturb=True), 'mid')
    wts = {k: v for k, v in wts_all.items() if k.startswith('update_block')}
    f1 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    f2 = rng.normal(size=(B, h, w, C)).astype(np.float32)
    net0 = np.tanh(rng.normal(size=(B, h, w, 128))).astype(np.float32)
    inp = np.maximum(rng.normal(size=(B, h, w, 128)), 0).astype(np.float32)
    flow_gt = (rng.normal(size=(B, 8 * h, 8 * w, 2)) * 2).astype(np.float32)
    valid = rng.uniform(size=(B, 8 * h, 8 * w)) < 0.9
    dev = CorrBlock(f1, f2, 4, 4)

    def run(params, tape_dtype):
        grad.clear_pack_cache()
        preds, tape = grad.loop_forward(params, dev, net0, inp, iters, tape_dtype=tape_dtype)
        d_preds = grad.sequence_loss_grad((flow_gt, valid), preds, gamma=0.8, max_flow=400)
        d_net0, d_inp, d_pyr, wg = grad.loop_backward(params, dev, tape, d_preds)
        return [_np(p) for p in preds], _np(d_net0), _np(d_inp), {k: _np(v) for k, v in wg.items()}, tape

    p_np, n_np, i_np, w_np, _ = run(wts, 'f32')
    dw = {k: _dev.to_device(np.ascontiguousarray(v)).as_subclass(torch.Tensor).clone() for k, v in wts.items()}
    p_dv, n_dv, i_dv, w_dv, _ = run(dw, 'f32')
    for a, b_ in zip(p_np, p_dv):
        np.testing.assert_array_equal(a, b_)
    np.testing.assert_array_equal(n_np, n_dv)
    np.testing.assert_array_equal(i_np, i_dv)
    for k in w_np:
        np.testing.assert_array_equal(w_np[k], w_dv[k])
    p_bf, n_bf, i_bf, w_bf, tape = run(dw, 'bf16')
    for a, b_ in zip(p_np, p_bf):
        np.testing.assert_array_equal(a, b_)                     # the forward itself is fp32
    stored = [v for t in tape for v in t['saved'].values() if isinstance(v, tuple)]
    assert len(stored) >= 10 * iters and all(v[1].dtype == torch.bfloat16 for v in stored)
    rel = lambda a, b_: float(np.abs(a - b_).max() / max(np.abs(b_).max(), 1e-30))
    rel2 = lambda a, b_: float(np.linalg.norm((a - b_).astype(np.float64)) / max(np.linalg.norm(b_.astype(np.float64)), 1e-30))
    worst_w = max(rel2(w_bf[k], w_np[k]) for k in w_np)
    report('bf16 tape vs fp32 tape', d_net0=rel(n_bf, n_np), d_inp=rel(i_bf, i_np), worst_weight_grad_l2=worst_w,
           tape_tensors_as_bf16=len(stored))
    assert rel(n_bf, n_np) <= 5e-3 and rel(i_bf, i_np) <= 5e-3
    assert worst_w <= 2e-2, {k: rel2(w_bf[k], w_np[k]) for k in w_np if rel2(w_bf[k], w_np[k]) > 5e-3}


def test_bf16_casts_round_to_nearest_even_and_dropout_is_a_scaled_mask(rng):
    from tf_raft_amd import _dev, grad
    x = np.concatenate([rng.normal(size=4096).astype(np.float32) * 100, np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, np.inf, -np.inf,
                                                                              3.3895314e38], np.float32)])
    xd = _dev.to_device(x).as_subclass(torch.Tensor)
    got = grad.to_bf16(xd)
    want = torch.tensor(x).to(torch.bfloat16)                    # torch's own conversion is round-to-nearest-even
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16))
    back = _np(grad.from_bf16(got))
    np.testing.assert_array_equal(back, want.to(torch.float32).numpy())
    # dropout: reproducible, ~rate of the elements zeroed, survivors scaled by 1 / (1 - rate), backward = the same mask
    v = _dev.to_device(np.ones((2, 16, 24, 64), np.float32)).as_subclass(torch.Tensor)
    y, mask = grad.dropout_forward(v, 0.25, seed=11)
    y2, _ = grad.dropout_forward(v, 0.25, seed=11)
    y3, _ = grad.dropout_forward(v, 0.25, seed=12)
    assert torch.equal(y, y2) and not torch.equal(y, y3)
    yn = _np(y)
    frac = float((yn == 0).mean())
    assert abs(frac - 0.25) < 0.01 and set(np.unique(yn)) == {0.0, np.float32(1.0 / 0.75)}
    dx = _np(grad.dropout_backward(2 * v, mask))
    np.testing.assert_array_equal(dx, 2 * yn)
    report('dropout', dropped_fraction=frac)


def test_train_step_runs_device_resident_with_dropout_and_bf16_tape(rng):
    """One full train_step with drop_rate > 0 (reference extractor.py:109-111: Dropout on the encoder outputs) and the bf16
    tape: finite loss, every trainable tensor moved, the master copies stay on the device (the NumPy dictionary is only
    refreshed on demand) and a second step continues from them."""
    import tf_raft_amd
    from tf_raft_amd import losses, training
    from tf_raft_amd import weights as wm
    wts = wm.condition_weights('raft', wm.init_weights('raft', seed=2, perturb=True), 'mid')
    model = tf_raft_amd.RAFT(drop_rate=0.1, weights=wts, iters=2, iters_pred=2)
    model.compile(optimizer=training.AdamW(1e-4, 1e-3), clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error,
                  tape_dtype='bf16')
    B, H, W = 2, 64, 96
    data = (rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32), rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32),
            (rng.normal(size=(B, H, W, 2)) * 2).astype(np.float32), np.ones((B, H, W), bool))
    r1 = model.train_step(data)
    assert model._host_stale and all(v.is_cuda for v in model._dw.values())
    assert model._weights['update_block/flow_head/conv1/kernel'] is not None     # the host dictionary still holds the OLD values ...
    np.testing.assert_array_equal(model._weights['fnet/conv1/kernel'], wts['fnet/conv1/kernel'])
    r2 = model.train_step(data)
    new = model.get_weights_dict()                                                # ... until somebody asks
    assert not model._host_stale
    assert np.isfinite(float(r1['loss'])) and np.isfinite(float(r2['loss']))
    moved = [k for k in new if not np.array_equal(new[k], wts[k])]
    assert len(moved) >= 154, len(moved)
    out = model([data[0], data[1]])                                               # inference re-packs from the trained weights
    assert np.isfinite(_np(out[-1])).all()
    report('train_step device-resident, dropout 0.1, bf16 tape', loss1=float(r1['loss']), loss2=float(r2['loss']), tensors_moved=len(moved))


@pytest.mark.parametrize('ksize,cin,cout', [((3, 3), 128, 64), ((3, 3), 36, 100), ((1, 5), 256, 128), ((5, 1), 40, 256), ((1, 1), 352, 256),
                                            ((7, 7), 4, 64)])
@pytest.mark.parametrize('dgrad', [False, True])
def test_fused_training_pack_equals_the_tensor_op_chain(rng, ksize, cin, cout, dgrad):
    """raft_pack_train_conv_f32 (one launch: optional flip / transpose for the input gradient, optional Winograd transform in
    float64, zero-padded operand layout) against the chain of torch ops it replaces, for a DEVICE kernel: direct layout, F(2x2, 3x3)
    and F(4, 5).  Same float64 arithmetic, possibly another summation order: equal to one fp32 ulp."""
    from tf_raft_amd import grad, packing
    kh, kw = ksize
    if dgrad and kh % 2 == 0:
        pytest.skip('input-gradient kernels are odd-sized')
    kernel = torch.as_tensor(rng.normal(size=(kh, kw, cin, cout)).astype(np.float32)).cuda()
    bias = torch.as_tensor(rng.normal(size=(cout,)).astype(np.float32)).cuda()
    k_ch = cout if dgrad else cin
    cpad = packing.round_up(k_ch, 32)
    for wino in (None, {(3, 3): '2d', (1, 5): '1d', (5, 1): '1d'}.get(ksize)):
        if wino is None and ksize in ((3, 3), (1, 5), (5, 1)) and cin == 36:
            continue
        kd = grad._dgrad_kernel_any(kernel) if dgrad else kernel
        want_wp, want_b, want_npad = grad._pack_conv_any(grad._wino_transform_any(kd) if wino else kd, None if dgrad else bias, [(k_ch, cpad)])
        got_wp, got_b, got_npad = grad._pack_train_device(kernel, bias, wino, dgrad, cpad)
        assert got_npad == want_npad and tuple(got_wp.shape) == tuple(want_wp.shape)
        np.testing.assert_allclose(got_wp.cpu().numpy(), want_wp.cpu().numpy(), rtol=2.5e-7, atol=1e-9)
        np.testing.assert_array_equal(got_b.cpu().numpy(), want_b.cpu().numpy())
        if wino is None:
            np.testing.assert_array_equal(got_wp.cpu().numpy(), want_wp.cpu().numpy())          # pure data movement: bit-equal
        if ksize not in ((3, 3), (1, 5), (5, 1)):
            break
