"""Oracle restatement of reference tf_raft/layers/update.py and extractor.py
(test infrastructure only).  Layers are plain functions of (weights dict, prefix, inputs);
weights are the Keras-layout arrays of ``tf_raft_amd.weights`` converted to torch."""
from __future__ import annotations

import torch

from . import tf_ops


class W:
    """Read-only view of a weight dict as torch tensors of one dtype."""

    def __init__(self, weights, dtype=torch.float32):
        self.dtype = dtype
        self.t = {k: torch.as_tensor(v).to(dtype) for k, v in weights.items()}

    def conv(self, name, x, stride=1, padding='same'):
        return tf_ops.conv2d(x, self.t[f'{name}/kernel'], self.t[f'{name}/bias'], stride, padding)

    def has(self, name):
        return name in self.t

    def __getitem__(self, name):
        return self.t[name]


# ---------------------------------------------------------------- update.py

def flow_head(w: W, p, x):
    """reference update.py:5-14."""
    return w.conv(f'{p}/conv2', torch.relu(w.conv(f'{p}/conv1', x)))


def conv_gru(w: W, p, h, x):
    """reference update.py:17-35."""
    hx = torch.cat([h, x], dim=-1)
    z = torch.sigmoid(w.conv(f'{p}/convz', hx))
    r = torch.sigmoid(w.conv(f'{p}/convr', hx))
    q = torch.tanh(w.conv(f'{p}/convq', torch.cat([r * h, x], dim=-1)))
    return (1 - z) * h + z * q


def sep_conv_gru(w: W, p, h, x):
    """reference update.py:38-67: horizontal (1x5) then vertical (5x1) pass."""
    for s in ('1', '2'):
        hx = torch.cat([h, x], dim=-1)
        z = torch.sigmoid(w.conv(f'{p}/convz{s}', hx))
        r = torch.sigmoid(w.conv(f'{p}/convr{s}', hx))
        q = torch.tanh(w.conv(f'{p}/convq{s}', torch.cat([r * h, x], dim=-1)))
        h = (1 - z) * h + z * q
    return h


def small_motion_encoder(w: W, p, flow, corr):
    """reference update.py:70-85."""
    cor = torch.relu(w.conv(f'{p}/convc1', corr))
    flo = torch.relu(w.conv(f'{p}/convf1', flow))
    flo = torch.relu(w.conv(f'{p}/convf2', flo))
    out = torch.relu(w.conv(f'{p}/conv', torch.cat([cor, flo], dim=-1)))
    return torch.cat([out, flow], dim=-1)


def basic_motion_encoder(w: W, p, flow, corr):
    """reference update.py:88-106 (convc1 is a 1x1 conv with Keras-default 'valid' padding)."""
    cor = torch.relu(w.conv(f'{p}/convc1', corr, padding='valid'))
    cor = torch.relu(w.conv(f'{p}/convc2', cor))
    flo = torch.relu(w.conv(f'{p}/convf1', flow))
    flo = torch.relu(w.conv(f'{p}/convf2', flo))
    out = torch.relu(w.conv(f'{p}/conv', torch.cat([cor, flo], dim=-1)))
    return torch.cat([out, flow], dim=-1)


def small_update_block(w: W, p, net, inp, corr, flow):
    """reference update.py:109-125."""
    motion = small_motion_encoder(w, f'{p}/encoder', flow, corr)
    inp = torch.cat([inp, motion], dim=-1)
    net = conv_gru(w, f'{p}/gru', net, inp)
    return net, None, flow_head(w, f'{p}/flow_head', net)


def basic_update_block(w: W, p, net, inp, corr, flow):
    """reference update.py:128-153."""
    motion = basic_motion_encoder(w, f'{p}/encoder', flow, corr)
    inp = torch.cat([inp, motion], dim=-1)
    net = sep_conv_gru(w, f'{p}/gru', net, inp)
    delta_flow = flow_head(w, f'{p}/flow_head', net)
    m = torch.relu(w.conv(f'{p}/mask/0', net))
    mask = 0.25 * w.conv(f'{p}/mask/2', m, padding='valid')
    return net, mask, delta_flow


# ---------------------------------------------------------------- extractor.py

def normalization(w: W, p, x, training):
    """reference extractor.py:6-16: dispatch on which parameters exist for this layer."""
    if w.has(f'{p}/moving_mean'):
        return tf_ops.batch_norm(x, w[f'{p}/gamma'], w[f'{p}/beta'],
                                 w[f'{p}/moving_mean'], w[f'{p}/moving_variance'], training)
    if w.has(f'{p}/gamma'):
        return tf_ops.instance_norm(x, w[f'{p}/gamma'], w[f'{p}/beta'])
    return x


def res_block(w: W, p, x, strides, training):
    """reference extractor.py:19-49."""
    fx = torch.relu(normalization(w, f'{p}/norm1', w.conv(f'{p}/conv1', x, strides), training))
    fx = torch.relu(normalization(w, f'{p}/norm2', w.conv(f'{p}/conv2', fx), training))
    if strides != 1:
        x = w.conv(f'{p}/downsample/0', x, strides, padding='valid')
        x = normalization(w, f'{p}/downsample/1', x, training)
    return torch.relu(x + fx)


def encoder(w: W, p, inputs, training=False):
    """reference extractor.py:113-130 (BasicEncoder.call) / 158-175 (SmallEncoder.call):
    identical structure, channel counts come from the weights."""
    is_list = isinstance(inputs, (tuple, list))
    x = torch.cat(list(inputs), dim=0) if is_list else inputs
    x = torch.relu(normalization(w, f'{p}/norm1', w.conv(f'{p}/conv1', x, 2), training))
    for li, s in ((1, 1), (2, 2), (3, 2)):
        x = res_block(w, f'{p}/layer{li}/0', x, s, training)
        x = res_block(w, f'{p}/layer{li}/1', x, 1, training)
    x = w.conv(f'{p}/conv2', x, padding='valid')
    if is_list:
        half = x.shape[0] // 2
        return x[:half], x[half:]
    return x
