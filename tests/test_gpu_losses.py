"""Evaluation-side reductions (csrc/metrics.hip behind tf_raft_amd.losses) against the reference's own known answers
(tests/losses/test_losses.py, committed in tests/golden/reference_known_answers.json) and the NumPy oracle.  GPU only.

Tolerances: the kernels accumulate in float64 and round once, the reference reduces in fp32 -- compared at rtol 1e-5
(sequence_loss, EPE); the rates u1 / u3 / u5 are ratios of integer counts and must agree to 1e-6 (a pixel whose EPE
sits within an ulp of 1 / 3 / 5 could flip between the two fp32 evaluations; the random inputs have none)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def known():
    with open(os.path.join(GOLDEN, 'reference_known_answers.json')) as f:
        return json.load(f)['losses']


def _fixture(k):
    flow_gt = (np.array(k['flow_gt_3x3x2_plus_0p1'], dtype=np.float64) - k['flow_gt_minus'])[None]
    valid = np.array(k['valid_3x3'])[None]
    preds = [np.zeros_like(flow_gt) for _ in range(k['n_predictions'])]
    return flow_gt, valid, preds


def test_reference_known_answers(known):
    """reference tests/losses/test_losses.py:27-67, same data, same expected values, same tolerances."""
    from tf_raft_amd import losses
    flow_gt, valid, preds = _fixture(known)
    want = 0.0
    for i, p in enumerate(preds):
        want += known['gamma'] ** (len(preds) - i - 1) * np.mean(valid[..., None] * np.abs(p - flow_gt))
    got = losses.sequence_loss((flow_gt, valid), preds, gamma=known['gamma'])
    np.testing.assert_almost_equal(float(got), want, decimal=6)
    info = losses.end_point_error([flow_gt, valid], preds[-1])
    np.testing.assert_almost_equal(float(info['epe']), np.mean(np.sqrt((np.arange(1, 9) - 0.1) ** 2)), decimal=2)
    for u in ('u1', 'u3', 'u5'):
        np.testing.assert_almost_equal(float(info[u]), known[u], decimal=2)
    assert info['epe'].numpy().dtype == np.float32 and info['epe'].numpy().shape == ()


def _random_case(rng, shape, n_pred, big=True):
    flow_gt = (rng.normal(size=shape + (2,)) * 4).astype(np.float32)
    if big:   # displacements beyond max_flow are excluded even where valid is set
        flow_gt.reshape(-1, 2)[:: 97] *= 200.0
    valid = rng.uniform(size=shape) < 0.8
    preds = [(flow_gt + rng.normal(size=shape + (2,)) * (3.0 / (i + 1))).astype(np.float32) for i in range(n_pred)]
    return flow_gt, valid, preds


@pytest.mark.parametrize('shape,n_pred', [((1, 3, 5), 1), ((2, 64, 96), 12), ((1, 7, 129), 24), ((4, 448, 512), 24)])
def test_losses_match_oracle(rng, shape, n_pred):
    from oracle import losses as oracle
    from tf_raft_amd import _dev, losses
    flow_gt, valid, preds = _random_case(rng, shape, n_pred)
    want_loss = oracle.sequence_loss((flow_gt, valid), preds, gamma=0.8, max_flow=400)
    want = oracle.end_point_error((flow_gt, valid), preds[-1], max_flow=400)
    # (a) the list as separate host arrays (gathered once), (b) as views of one device buffer (read in place)
    got_a = float(losses.sequence_loss((flow_gt, valid), preds, gamma=0.8, max_flow=400))
    buf = _dev.to_device(np.stack(preds))
    views = [buf[i] for i in range(n_pred)]
    got_b = float(losses.sequence_loss((_dev.to_device(flow_gt), torch.as_tensor(valid).cuda()), views, gamma=0.8))
    assert got_a == got_b
    np.testing.assert_allclose(got_a, want_loss, rtol=1e-5)
    info = losses.end_point_error((flow_gt, valid), views[-1])
    report(f'losses {shape} x{n_pred}', loss=got_a, loss_oracle=float(want_loss), epe=float(info['epe']), epe_oracle=want['epe'])
    np.testing.assert_allclose(float(info['epe']), want['epe'], rtol=1e-5)
    for u in ('u1', 'u3', 'u5'):
        np.testing.assert_allclose(float(info[u]), want[u], atol=1e-6)
    # deterministic: same bits on a second evaluation
    again = losses.end_point_error((flow_gt, valid), views[-1])
    assert all(float(again[k]) == float(info[k]) for k in info)
    assert float(losses.sequence_loss((flow_gt, valid), views, gamma=0.8)) == got_b


def test_loss_properties_at_full_size(rng):
    """Size-independent properties at the benchmark shape: scaling the residual scales the loss, a perfect prediction
    gives zero loss / zero EPE / all rates one, gamma = 1 makes the loss the sum of the per-prediction means."""
    from tf_raft_amd import _dev, losses
    shape, n = (4, 448, 512), 24
    flow_gt = _dev.to_device((rng.normal(size=shape + (2,)) * 4).astype(np.float32))
    valid = torch.ones(shape, dtype=torch.bool, device='cuda')
    res = _dev.to_device(rng.normal(size=(n,) + shape + (2,)).astype(np.float32))
    preds = [flow_gt + res[i] for i in range(n)]
    preds2 = [flow_gt + 2.0 * res[i] for i in range(n)]      # |2 r| = 2 |r| exactly up to the rounding of gt + r
    l1 = float(losses.sequence_loss((flow_gt, valid), preds))
    l2 = float(losses.sequence_loss((flow_gt, valid), preds2))
    np.testing.assert_allclose(l2, 2.0 * l1, rtol=1e-5)
    perfect = [flow_gt.clone() for _ in range(3)]
    assert float(losses.sequence_loss((flow_gt, valid), perfect)) == 0.0
    info = losses.end_point_error((flow_gt, valid), perfect[-1])
    assert float(info['epe']) == 0.0 and float(info['u1']) == 1.0 and float(info['u5']) == 1.0
    per = [float(losses.sequence_loss((flow_gt, valid), [p], gamma=1.0)) for p in preds[:4]]
    np.testing.assert_allclose(float(losses.sequence_loss((flow_gt, valid), preds[:4], gamma=1.0)), sum(per), rtol=1e-6)


def test_edge_cases(rng):
    from tf_raft_amd import losses
    flow_gt = rng.normal(size=(1, 4, 6, 2)).astype(np.float32)
    pred = np.zeros_like(flow_gt)
    # no valid pixel: the reference takes the mean of an empty tensor (NaN); the loss is a mean over all pixels (0)
    none = np.zeros((1, 4, 6), bool)
    info = losses.end_point_error((flow_gt, none), pred)
    assert np.isnan(float(info['epe'])) and np.isnan(float(info['u3']))
    assert float(losses.sequence_loss((flow_gt, none), [pred, pred])) == 0.0
    # everything beyond max_flow
    assert float(losses.sequence_loss((np.full_like(flow_gt, 1e4), ~none), [pred], max_flow=400)) == 0.0
    # an empty prediction list: flow_loss = 0.0 (losses.py:8)
    assert float(losses.sequence_loss((flow_gt, ~none), [])) == 0.0
    # uint8 / float masks are accepted like bool ones
    a = float(losses.end_point_error((flow_gt, (~none).astype(np.uint8)), pred)['epe'])
    b = float(losses.end_point_error((flow_gt, ~none), pred)['epe'])
    assert a == b
    with pytest.raises(ValueError):
        losses.end_point_error((flow_gt, none[:, :2]), pred)
    with pytest.raises(ValueError):
        losses.end_point_error((flow_gt, none), pred[:, :2])
    with pytest.raises(ValueError):
        losses.sequence_loss((flow_gt[..., :1], none), [pred])
    with pytest.raises(ValueError):
        losses.sequence_loss(flow_gt, [pred])
    with pytest.raises(ValueError):
        losses.sequence_loss((flow_gt, none), [pred] * 65)


def test_metrics_c_abi_argument_checks():
    from tf_raft_amd import _dev
    lib = _dev.lib()
    x = torch.zeros(64, device='cuda')
    v = torch.zeros(32, dtype=torch.uint8, device='cuda')
    ws = torch.zeros(int(lib.raft_metrics_workspace_doubles()), dtype=torch.float64, device='cuda')
    assert lib.raft_flow_metrics_f32(None, _dev.ptr(v), _dev.ptr(x), 32, 400.0, _dev.ptr(x), _dev.ptr(ws), None) < 0
    assert lib.raft_flow_metrics_f32(_dev.ptr(x), _dev.ptr(v), _dev.ptr(x), 0, 400.0, _dev.ptr(x), _dev.ptr(ws), None) < 0
    assert lib.raft_sequence_loss_f32(_dev.ptr(x), _dev.ptr(v), _dev.ptr(x), 64, 65, 32, 0.8, 400.0, _dev.ptr(x), _dev.ptr(ws), None) < 0
    assert lib.raft_sequence_loss_f32(_dev.ptr(x), _dev.ptr(v), _dev.ptr(x), 32, 1, 32, 0.8, 400.0, _dev.ptr(x), _dev.ptr(ws), None) < 0
    torch.cuda.synchronize()


def test_end_point_error_metric_and_test_step(rng):
    """EndPointError (losses.py:46-90) and RAFT.compile / test_step / reset_metrics (model.py:111-170) against the
    oracle's end_point_error of the model's own final prediction."""
    from oracle import losses as oracle
    from tf_raft_amd import losses
    from tf_raft_amd.model import RAFT
    B, H, W = 2, 64, 96
    model = RAFT(iters_pred=3, seed=5)
    with pytest.raises(RuntimeError):
        model.test_step((None, None, None, None))
    model.compile(optimizer=None, clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error)
    assert list(model.flow_metrics) == ['loss', 'epe', 'u1', 'u3', 'u5']
    metric = losses.EndPointError(max_flow=400)
    assert np.isnan(metric.result()['epe'])
    sums = {k: 0.0 for k in ('epe', 'u1', 'u3', 'u5')}
    out = None
    for step in range(2):
        im1 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
        im2 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
        flow = (rng.normal(size=(B, H, W, 2)) * 2).astype(np.float32)
        valid = rng.uniform(size=(B, H, W)) < 0.9
        preds = model([im1, im2], training=False)
        want = oracle.end_point_error((flow, valid), preds[-1].numpy())
        for k in sums:
            sums[k] += want[k]
        out = model.test_step((im1, im2, flow, valid))
        metric.update_state((flow, valid), preds)
        for k in sums:
            np.testing.assert_allclose(out[k], sums[k] / (step + 1), rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(float(metric.result()[k]), sums[k] / (step + 1), rtol=2e-5, atol=1e-6)
    assert out['loss'] == 0.0                      # only train_step feeds it
    model.reset_metrics()
    assert all(m.count == 0 for m in model.flow_metrics.values())
    fresh = RAFT(iters_pred=2)
    with pytest.raises(RuntimeError):              # keras raises for an un-compiled model too
        fresh.train_step((im1, im2, flow, valid))


def test_sequence_loss_multiplies_by_the_mask_like_the_reference(rng):
    """reference losses.py:14-19: ``mean(valid * |pred - gt|)`` -- the mask MULTIPLIES, so a NaN / Inf prediction at a
    masked pixel still poisons the loss (0 * NaN = NaN), while end_point_error (losses.py:32, boolean selection
    ``epe[valid]``) ignores it.  Both behaviours are the reference's and both are reproduced."""
    from oracle import losses as oracle
    from tf_raft_amd import losses
    flow_gt, valid, preds = _random_case(rng, (1, 16, 24), 3, big=False)
    valid[0, 3, 5] = False
    clean = float(losses.sequence_loss((flow_gt, valid), preds))
    assert np.isfinite(clean)
    np.testing.assert_allclose(clean, oracle.sequence_loss((flow_gt, valid), preds), rtol=1e-5)
    bad = [p.copy() for p in preds]
    bad[1][0, 3, 5, 0] = np.nan                              # masked pixel
    assert np.isnan(float(losses.sequence_loss((flow_gt, valid), bad)))
    assert np.isnan(oracle.sequence_loss((flow_gt, valid), bad))
    bad[1][0, 3, 5, 0] = np.inf
    assert np.isnan(float(losses.sequence_loss((flow_gt, valid), bad)))        # 0 * inf
    info = losses.end_point_error([flow_gt, valid], bad[1])
    assert np.isfinite(float(info['epe']))                   # selection, not multiplication
