"""Weight container for RAFT / SmallRAFT in **Keras layout**.

The reference creates its weights lazily inside Keras layers
(reference tf_raft/layers/update.py:5-153, extractor.py:19-49, 88-175); "random
weights" for an un-trained ``RAFT()`` therefore means the Keras defaults:
``glorot_uniform`` conv kernels ``(kh, kw, Cin, Cout)``, zero biases, instance-norm
``gamma=1, beta=0`` and batch-norm ``gamma=1, beta=0, moving_mean=0, moving_variance=1``.

This module describes every layer of both models as a flat table and generates
such weights from a seeded NumPy generator.  The same dict of arrays feeds the
device path (``tf_raft_amd.model``) and the CPU oracle (``oracle/``), so parity
tests compare like with like.  ``perturb=True`` additionally randomises biases and
norm parameters so that tests exercise those code paths (the Keras defaults make
all of them identity/zero).

Names follow the attribute paths of the reference layers, e.g.
``fnet/layer2/0/downsample/0/kernel`` or ``update_block/gru/convz1/kernel``.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# layer tables
# ----------------------------------------------------------------------------
# Each entry: (name, kind, shape-info)
#   kind 'conv'  -> (kh, kw, cin, cout)
#   kind 'in'    -> channels   (instance norm: gamma, beta)
#   kind 'bn'    -> channels   (batch norm: gamma, beta, moving_mean, moving_variance)

ENCODER_DIMS = {
    # reference extractor.py:95-102 (BasicEncoder) and 140-147 (SmallEncoder)
    'basic': (64, 64, 96, 128),
    'small': (32, 32, 64, 96),
}


def _norm_entries(prefix: str, norm_type, ch: int):
    if norm_type == 'instance':
        return [(prefix, 'in', ch)]
    if norm_type == 'batch':
        return [(prefix, 'bn', ch)]
    if norm_type is None:
        return []
    raise ValueError(f'Invalid norm_type specified: {norm_type}')


def _resblock_entries(prefix: str, cin: int, filters: int, norm_type, strides: int):
    # reference extractor.py:19-39
    e = [(f'{prefix}/conv1', 'conv', (3, 3, cin, filters)),
         (f'{prefix}/conv2', 'conv', (3, 3, filters, filters))]
    e += _norm_entries(f'{prefix}/norm1', norm_type, filters)
    e += _norm_entries(f'{prefix}/norm2', norm_type, filters)
    if strides != 1:
        e.append((f'{prefix}/downsample/0', 'conv', (1, 1, cin, filters)))
        e += _norm_entries(f'{prefix}/downsample/1', norm_type, filters)
    return e


def encoder_entries(prefix: str, variant: str, norm_type, output_dim: int):
    """reference extractor.py:88-111 / 133-156."""
    c0, c1, c2, c3 = ENCODER_DIMS[variant]
    e = [(f'{prefix}/conv1', 'conv', (7, 7, 3, c0))]
    e += _norm_entries(f'{prefix}/norm1', norm_type, c0)
    cin = c0
    for li, (f, s) in enumerate(((c1, 1), (c2, 2), (c3, 2)), start=1):
        e += _resblock_entries(f'{prefix}/layer{li}/0', cin, f, norm_type, s)
        e += _resblock_entries(f'{prefix}/layer{li}/1', f, f, norm_type, 1)
        cin = f
    e.append((f'{prefix}/conv2', 'conv', (1, 1, c3, output_dim)))
    return e


def basic_update_entries(prefix='update_block', hdim=128, cdim=128, corr_ch=324):
    """reference update.py:88-106 (BasicMotionEncoder), 38-49 (SepConvGRU),
    5-11 (FlowHead), 128-141 (BasicUpdateBlock)."""
    gin = hdim + cdim + 128          # hx = [h | inp | motion(126) | flow(2)]
    return [
        (f'{prefix}/encoder/convc1', 'conv', (1, 1, corr_ch, 256)),
        (f'{prefix}/encoder/convc2', 'conv', (3, 3, 256, 192)),
        (f'{prefix}/encoder/convf1', 'conv', (7, 7, 2, 128)),
        (f'{prefix}/encoder/convf2', 'conv', (3, 3, 128, 64)),
        (f'{prefix}/encoder/conv', 'conv', (3, 3, 192 + 64, 128 - 2)),
        (f'{prefix}/gru/convz1', 'conv', (1, 5, gin, hdim)),
        (f'{prefix}/gru/convr1', 'conv', (1, 5, gin, hdim)),
        (f'{prefix}/gru/convq1', 'conv', (1, 5, gin, hdim)),
        (f'{prefix}/gru/convz2', 'conv', (5, 1, gin, hdim)),
        (f'{prefix}/gru/convr2', 'conv', (5, 1, gin, hdim)),
        (f'{prefix}/gru/convq2', 'conv', (5, 1, gin, hdim)),
        (f'{prefix}/flow_head/conv1', 'conv', (3, 3, hdim, 256)),
        (f'{prefix}/flow_head/conv2', 'conv', (3, 3, 256, 2)),
        (f'{prefix}/mask/0', 'conv', (3, 3, hdim, 256)),
        (f'{prefix}/mask/2', 'conv', (1, 1, 256, 64 * 9)),
    ]


def small_update_entries(prefix='update_block', hdim=96, cdim=64, corr_ch=196):
    """reference update.py:70-85 (SmallMotionEncoder), 17-24 (ConvGRU),
    109-116 (SmallUpdateBlock, FlowHead(128))."""
    gin = hdim + cdim + 82           # hx = [h | inp | motion(80) | flow(2)]
    return [
        (f'{prefix}/encoder/convc1', 'conv', (1, 1, corr_ch, 96)),
        (f'{prefix}/encoder/convf1', 'conv', (7, 7, 2, 64)),
        (f'{prefix}/encoder/convf2', 'conv', (3, 3, 64, 32)),
        (f'{prefix}/encoder/conv', 'conv', (3, 3, 96 + 32, 80)),
        (f'{prefix}/gru/convz', 'conv', (3, 3, gin, hdim)),
        (f'{prefix}/gru/convr', 'conv', (3, 3, gin, hdim)),
        (f'{prefix}/gru/convq', 'conv', (3, 3, gin, hdim)),
        (f'{prefix}/flow_head/conv1', 'conv', (3, 3, hdim, 128)),
        (f'{prefix}/flow_head/conv2', 'conv', (3, 3, 128, 2)),
    ]


def model_entries(variant: str):
    """Full layer table of RAFT (reference model.py:10-30) or SmallRAFT (173-188)."""
    if variant == 'raft':
        return (encoder_entries('fnet', 'basic', 'instance', 256)
                + encoder_entries('cnet', 'basic', 'batch', 128 + 128)
                + basic_update_entries())
    if variant == 'small':
        return (encoder_entries('fnet', 'small', 'instance', 128)
                + encoder_entries('cnet', 'small', None, 96 + 64)
                + small_update_entries())
    raise ValueError(f'unknown model variant {variant!r}')


# ----------------------------------------------------------------------------
# generation
# ----------------------------------------------------------------------------

def _glorot_uniform(rng: np.random.Generator, shape: Tuple[int, int, int, int]):
    kh, kw, cin, cout = shape
    fan_in, fan_out = kh * kw * cin, kh * kw * cout
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def init_weights(variant: str, seed: int = 0, perturb: bool = False) -> 'OrderedDict[str, np.ndarray]':
    """Seeded Keras-default weights for ``variant`` in {'raft', 'small'}.

    perturb=True: biases ~ U(-0.1, 0.1), gamma ~ U(0.5, 1.5), beta ~ U(-0.2, 0.2),
    moving_mean ~ U(-0.2, 0.2), moving_variance ~ U(0.5, 1.5) -- used by tests so
    that bias / norm-parameter handling is actually exercised.
    """
    rng = np.random.default_rng(seed)
    w: 'OrderedDict[str, np.ndarray]' = OrderedDict()

    def vec(ch, lo, hi, default):
        if perturb:
            return rng.uniform(lo, hi, size=(ch,)).astype(np.float32)
        return np.full((ch,), default, dtype=np.float32)

    for name, kind, info in model_entries(variant):
        if kind == 'conv':
            w[f'{name}/kernel'] = _glorot_uniform(rng, info)
            w[f'{name}/bias'] = vec(info[3], -0.1, 0.1, 0.0)
        elif kind == 'in':
            w[f'{name}/gamma'] = vec(info, 0.5, 1.5, 1.0)
            w[f'{name}/beta'] = vec(info, -0.2, 0.2, 0.0)
        elif kind == 'bn':
            w[f'{name}/gamma'] = vec(info, 0.5, 1.5, 1.0)
            w[f'{name}/beta'] = vec(info, -0.2, 0.2, 0.0)
            w[f'{name}/moving_mean'] = vec(info, -0.2, 0.2, 0.0)
            w[f'{name}/moving_variance'] = vec(info, 0.5, 1.5, 1.0)
        else:  # pragma: no cover
            raise AssertionError(kind)
    return w


# Contractive flow-head regime for free-running parity tests (tests/golden/make_conditioning.py).
# With Keras-default weights an untrained RAFT's flow grows ~7 px per iteration, taps cross the sampler's clamp
# discontinuity (reference corr.py:41-48: a clamped coordinate is an integer and samples 0) and ANY two fp32
# evaluations of the recurrence -- the oracle in fp32 and in fp64 included -- part ways after ~10 iterations.
# Scaling flow_head.conv2 and giving it a small positive bias keeps every coordinate x + flow strictly inside
# (x, x + 1) for the whole loop (flow in ~[0.1, 0.9] px after 24 iterations), so no tap can flip and the 1e-3 bound of
# BASELINE.json's north_star is testable on flow_predictions[-1] itself.  Every other layer keeps its default weights.
CONDITIONED_HEAD = {'raft': (0.01, (0.02, 0.018)), 'small': (0.005, (0.02, 0.018))}
# Mid regime (round 3): the same construction with a flow head that is 20x stronger and drifts (+0.2, -0.15) feature pixels per
# iteration -- after 24 iterations the low-resolution flow spans about [0, 9] x [-10, 2] px with a standard deviation of ~1 px, so
# lookup taps cross integers at every pyramid level, windows slide over the clamped borders and the coarse levels are sampled
# away from the identity.  The recurrence is then only MOSTLY well conditioned: in about one run out of three a tap passes
# within rounding distance of a clamp boundary and the oracle in fp32 and in fp64 part ways on a few percent of the pixels
# (tests/golden/conditioning.json records which seeds and when), so the tests built on it assert the 1e-3 bound up to such a
# flip and require the flip to be local (tests/test_gpu_model.py::test_mid_regime_free_running).
MID_HEAD = {'raft': (0.2, (0.2, -0.15)), 'small': (0.1, (0.2, -0.15))}


# Jump regime (round 4): the conditioned regime plus an INTEGER drift per iteration.  The low-resolution flow after k
# iterations is k * jump + c_k with c_k the conditioned regime's small trajectory (strictly inside (0, 1) per axis), so every
# tap coordinate (x + flow) / 2^l + d keeps a fractional part >= ~0.01 / 2^l away from an integer -- no tap can sit within rounding
# distance of the sampler's discontinuities (reference corr.py:41-60: exact integers and clamped coordinates sample 0) -- while
# its integer part moves every iteration: lookup windows cross integers at every pyramid level, slide over the clamped borders
# (|flow| reaches 24 .. 48 low-resolution pixels on a 56 x 64 map) and the coarse levels are sampled far from the identity.
# tests/golden/make_conditioning.py records the oracle's fp32-vs-fp64 agreement AND the measured margin of every case.
JUMP_HEAD = {'raft': (0.01, (0.02, 0.018)), 'small': (0.005, (0.02, 0.018))}
JUMPS = ((1, -1), (-1, 1), (2, 1), (-1, -2), (1, 0), (0, -1), (-2, 2), (1, 1))


def condition_weights(variant: str, weights: Dict[str, np.ndarray], regime: str = 'conditioned', jump=None) -> 'OrderedDict[str, np.ndarray]':
    """Copy of ``weights`` with ``update_block/flow_head/conv2`` scaled / biased per ``CONDITIONED_HEAD`` (regime
    'conditioned'), ``MID_HEAD`` (regime 'mid') or ``JUMP_HEAD`` + the integer drift ``jump`` = (jx, jy) low-resolution pixels per
    iteration (regime 'jump'; 'jumpN' = entry N of ``JUMPS``)."""
    if regime.startswith('jump'):
        if jump is None:
            jump = JUMPS[int(regime[4:] or 0) % len(JUMPS)]
        scale, bias = JUMP_HEAD[variant]
        bias = (bias[0] + int(jump[0]), bias[1] + int(jump[1]))
    elif regime in ('conditioned', 'mid'):
        scale, bias = (CONDITIONED_HEAD if regime == 'conditioned' else MID_HEAD)[variant]
    else:
        raise ValueError(f"regime must be 'conditioned', 'mid' or 'jump[N]', got {regime!r}")
    w = OrderedDict(weights)
    k = 'update_block/flow_head/conv2/'
    w[k + 'kernel'] = (weights[k + 'kernel'] * np.float32(scale)).astype(np.float32)
    w[k + 'bias'] = np.asarray(bias, dtype=np.float32)
    return w


def count_params(weights: Dict[str, np.ndarray], prefix: str = '') -> int:
    return int(sum(v.size for k, v in weights.items() if k.startswith(prefix)))


def check_weights(variant: str, weights: Dict[str, np.ndarray]) -> None:
    """Raise ValueError when ``weights`` does not match the layer table of ``variant``."""
    expected: List[Tuple[str, Tuple[int, ...]]] = []
    for name, kind, info in model_entries(variant):
        if kind == 'conv':
            expected += [(f'{name}/kernel', tuple(info)), (f'{name}/bias', (info[3],))]
        elif kind == 'in':
            expected += [(f'{name}/gamma', (info,)), (f'{name}/beta', (info,))]
        else:
            expected += [(f'{name}/{p}', (info,)) for p in
                         ('gamma', 'beta', 'moving_mean', 'moving_variance')]
    for key, shape in expected:
        if key not in weights:
            raise ValueError(f'missing weight {key!r}')
        if tuple(weights[key].shape) != shape:
            raise ValueError(f'weight {key!r} has shape {tuple(weights[key].shape)}, expected {shape}')


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k.replace('/', '|'): v for k, v in weights.items()})


def load_weights(path: str) -> 'OrderedDict[str, np.ndarray]':
    with np.load(path) as z:
        return OrderedDict((k.replace('|', '/'), z[k].astype(np.float32)) for k in z.files)
