"""Second, independent CPU restatement of the producer: the classifier as a graph of stock
``torch.nn`` modules.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

PARITY UNPINNED (as ``oracle/xception_ref.py``): Slideflow / Keras cannot be installed here and the
reference holds no golden vector for this side (``requirements.txt:1,5``; call sites
``biscuit/experiment.py:917-922``, ``results.py:250-258``).  This file exists so that ONE hand's
mistake in the first restatement cannot hide: it shares no code with ``xception_ref.py`` and differs
from it in every mechanism --

* the network is written down as the list of Keras layer NAMES that ``keras.applications.Xception(
  include_top=False, pooling='avg')`` prints in ``model.summary()`` (block1_conv1, block1_conv1_bn,
  block1_conv1_act, ..., conv2d, batch_normalization, ..., add_11, ..., avg_pool), each with the
  names of its inbound layers, followed by Slideflow's head (``hp.py:11-13,21``: dropout 0.1, two
  hidden Dense(1024, relu), softmax over 2 classes) -- not as nested Python loops;
* every layer is a stock module: ``nn.Conv2d`` (``groups=C`` for the depthwise half), ``nn.BatchNorm2d(
  eps=1e-3)`` in eval mode with the four Keras vectors loaded UNFOLDED, ``nn.MaxPool2d(3, 2)`` on an
  explicitly padded input, ``nn.ReLU``, ``nn.Linear``;
* TensorFlow 'same' padding is computed per layer from the published formula
  ``total = max((ceil(n/s) - 1) * s + k - n, 0); before = total // 2`` and applied with
  ``nn.ZeroPad2d`` / ``nn.ConstantPad2d(-inf)``, so strided and pooled layers get their ASYMMETRIC pads
  from the rule, not from a table;
* ``tf.image.per_image_standardization`` (``results.py:256``) is restated in float32 numpy with the
  documented ``adjusted_stddev = max(stddev, 1/sqrt(N))``.

The dropout masks are this build's own Philox contract (``oracle/philox.py``): TensorFlow's stream
cannot be reproduced without TensorFlow, so both restatements draw the same masks by design.
"""
import math

import numpy as np
import torch
from torch import nn

from . import philox

KERAS_BN_EPS = 1e-3


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class _SamePad(nn.Module):
    """TensorFlow 'same' padding for a k x k window at stride s (value 0 for convolutions, -inf for max-pool)."""

    def __init__(self, k, s, value=0.0):
        super().__init__()
        self.k, self.s, self.value = k, s, value

    def forward(self, x):
        (t, b), (l, r) = _same_pad(x.shape[2], self.k, self.s), _same_pad(x.shape[3], self.k, self.s)
        return nn.functional.pad(x, (l, r, t, b), value=self.value)


class _Separable(nn.Module):
    """keras.layers.SeparableConv2D(filters, 3, padding='same', use_bias=False): depthwise 3x3
    (depth_multiplier 1) then pointwise 1x1, no activation or normalisation in between."""

    def __init__(self, cin, cout):
        super().__init__()
        self.pad = _SamePad(3, 1)
        self.depthwise = nn.Conv2d(cin, cin, 3, groups=cin, bias=False)
        self.pointwise = nn.Conv2d(cin, cout, 1, bias=False)

    def forward(self, x):
        return self.pointwise(self.depthwise(self.pad(x)))


def keras_xception_layers():
    """[(keras layer name, kind, args, inbound names)] in ``model.layers`` order."""
    L = []

    def add(name, kind, args, inbound):
        L.append((name, kind, args, inbound if isinstance(inbound, (list, tuple)) else [inbound]))
        return name

    x = add('block1_conv1', 'conv', (3, 32, 3, 2, 'valid'), 'input')
    x = add('block1_conv1_bn', 'bn', (32,), x)
    x = add('block1_conv1_act', 'relu', (), x)
    x = add('block1_conv2', 'conv', (32, 64, 3, 1, 'valid'), x)
    x = add('block1_conv2_bn', 'bn', (64,), x)
    x = add('block1_conv2_act', 'relu', (), x)
    n_conv, n_bn, n_add = 0, 0, 0

    def auto(base, n):
        return base if n == 0 else f'{base}_{n}'

    cin = 64
    # entry flow: the residual Conv2D / BatchNormalization carry Keras' automatic names
    for block, cout in ((2, 128), (3, 256), (4, 728)):
        r = add(auto('conv2d', n_conv), 'conv', (cin, cout, 1, 2, 'same'), x); n_conv += 1
        r = add(auto('batch_normalization', n_bn), 'bn', (cout,), r); n_bn += 1
        y = x
        if block != 2:
            y = add(f'block{block}_sepconv1_act', 'relu', (), y)
        y = add(f'block{block}_sepconv1', 'sep', (cin, cout), y)
        y = add(f'block{block}_sepconv1_bn', 'bn', (cout,), y)
        y = add(f'block{block}_sepconv2_act', 'relu', (), y)
        y = add(f'block{block}_sepconv2', 'sep', (cout, cout), y)
        y = add(f'block{block}_sepconv2_bn', 'bn', (cout,), y)
        y = add(f'block{block}_pool', 'maxpool', (), y)
        x = add(auto('add', n_add), 'add', (), [y, r]); n_add += 1
        cin = cout
    for block in range(5, 13):
        y = x
        for i in (1, 2, 3):
            y = add(f'block{block}_sepconv{i}_act', 'relu', (), y)
            y = add(f'block{block}_sepconv{i}', 'sep', (728, 728), y)
            y = add(f'block{block}_sepconv{i}_bn', 'bn', (728,), y)
        x = add(auto('add', n_add), 'add', (), [y, x]); n_add += 1
    r = add(auto('conv2d', n_conv), 'conv', (728, 1024, 1, 2, 'same'), x); n_conv += 1
    r = add(auto('batch_normalization', n_bn), 'bn', (1024,), r); n_bn += 1
    y = add('block13_sepconv1_act', 'relu', (), x)
    y = add('block13_sepconv1', 'sep', (728, 728), y)
    y = add('block13_sepconv1_bn', 'bn', (728,), y)
    y = add('block13_sepconv2_act', 'relu', (), y)
    y = add('block13_sepconv2', 'sep', (728, 1024), y)
    y = add('block13_sepconv2_bn', 'bn', (1024,), y)
    y = add('block13_pool', 'maxpool', (), y)
    x = add(auto('add', n_add), 'add', (), [y, r]); n_add += 1
    x = add('block14_sepconv1', 'sep', (1024, 1536), x)
    x = add('block14_sepconv1_bn', 'bn', (1536,), x)
    x = add('block14_sepconv1_act', 'relu', (), x)
    x = add('block14_sepconv2', 'sep', (1536, 2048), x)
    x = add('block14_sepconv2_bn', 'bn', (2048,), x)
    x = add('block14_sepconv2_act', 'relu', (), x)
    add('avg_pool', 'gap', (), x)
    return L


# Keras' automatic names of the four residual branches -> the names of this build's canonical weight dict
_RESIDUAL_BLOCK = {0: 2, 1: 3, 2: 4, 3: 13}


def canonical_weight_name(keras_name):
    if keras_name.startswith('conv2d'):
        k = int(keras_name[7:] or 0)
        return f'block{_RESIDUAL_BLOCK[k]}_res_conv'
    if keras_name.startswith('batch_normalization'):
        k = int(keras_name[20:] or 0)
        return f'block{_RESIDUAL_BLOCK[k]}_res_bn'
    return keras_name


class XceptionNN(nn.Module):
    """The hp.nature2022 classifier (``biscuit/hp.py:3-23``) as a graph of torch.nn modules."""

    def __init__(self, weights, dropout=0.1):
        super().__init__()
        self.rate = float(dropout)
        self.layers = keras_xception_layers()
        self.mods = nn.ModuleDict()
        for name, kind, args, _ in self.layers:
            if kind == 'conv':
                cin, cout, k, s, padding = args
                conv = nn.Conv2d(cin, cout, k, stride=s, bias=False)
                self.mods[name] = nn.Sequential(_SamePad(k, s), conv) if padding == 'same' else nn.Sequential(conv)
            elif kind == 'sep':
                self.mods[name] = _Separable(*args)
            elif kind == 'bn':
                self.mods[name] = nn.BatchNorm2d(args[0], eps=KERAS_BN_EPS)
            elif kind == 'relu':
                self.mods[name] = nn.ReLU()
            elif kind == 'maxpool':
                self.mods[name] = nn.Sequential(_SamePad(3, 2, float('-inf')), nn.MaxPool2d(3, 2))
        self.hidden_0 = nn.Linear(2048, 1024)
        self.hidden_1 = nn.Linear(1024, 1024)
        self.logits = nn.Linear(1024, 2)
        self._load(weights)
        self.eval()                                   # BatchNorm on its moving statistics (training=False)
        for p in self.parameters():
            p.requires_grad_(False)

    def _load(self, w):
        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        with torch.no_grad():
            for name, kind, args, _ in self.layers:
                cn = canonical_weight_name(name)
                m = self.mods[name] if name in self.mods else None
                if kind == 'conv':
                    m[-1].weight.copy_(t(w[cn + '/kernel']).permute(3, 2, 0, 1))        # HWIO -> OIHW
                elif kind == 'sep':
                    m.depthwise.weight.copy_(t(w[cn + '/depthwise_kernel']).permute(2, 3, 0, 1))   # [3,3,C,1] -> [C,1,3,3]
                    m.pointwise.weight.copy_(t(w[cn + '/pointwise_kernel']).permute(3, 2, 0, 1))
                elif kind == 'bn':
                    m.weight.copy_(t(w[cn + '/gamma'])); m.bias.copy_(t(w[cn + '/beta']))
                    m.running_mean.copy_(t(w[cn + '/moving_mean'])); m.running_var.copy_(t(w[cn + '/moving_variance']))
            for lin, nm in ((self.hidden_0, 'hidden_0'), (self.hidden_1, 'hidden_1'), (self.logits, 'logits')):
                lin.weight.copy_(t(w[nm + '/kernel']).t()); lin.bias.copy_(t(w[nm + '/bias']))

    @torch.no_grad()
    def features(self, x, keep=None):
        """x: float32 NCHW standardised tiles -> [n,2048].  ``keep`` (dict) receives every layer output by Keras name."""
        val = {'input': x}
        for name, kind, args, inbound in self.layers:
            a = val[inbound[0]]
            if kind == 'add':
                out = a + val[inbound[1]]
            elif kind == 'gap':
                out = a.mean(dim=(2, 3))
            else:
                out = self.mods[name](a)
            val[name] = out
            if keep is not None:
                keep[name] = out
        return val['avg_pool']

    @torch.no_grad()
    def head_pass(self, feat, tile_index, mc_pass, seed):
        """dropout -> hidden_0 -> relu -> dropout -> hidden_1 -> relu -> dropout -> logits -> softmax, dropout always on."""
        scale = np.float32(philox.dropout_scale(self.rate))
        h = feat
        for layer, (lin, width) in enumerate(((self.hidden_0, 2048), (self.hidden_1, 1024), (self.logits, 1024))):
            mask = philox.dropout_keep(seed, tile_index, mc_pass, layer, width, self.rate)
            h = lin(h * torch.from_numpy(mask.astype(np.float32)) * scale)
            if layer < 2:
                h = torch.relu(h)
        return torch.softmax(h, dim=1)


def per_image_standardization(tiles_u8):
    """tf.image.per_image_standardization on uint8 NHWC tiles -> float32 NCHW (results.py:256)."""
    x = np.asarray(tiles_u8).astype(np.float32)
    n = x.shape[0]
    num = x[0].size
    out = np.empty_like(x)
    for i in range(n):
        im = x[i].astype(np.float64)
        mean = im.mean()
        std = math.sqrt(((im - mean) ** 2).mean())
        adj = max(std, 1.0 / math.sqrt(num))
        out[i] = (x[i] - np.float32(mean)) * np.float32(1.0 / adj)
    return torch.from_numpy(out.transpose(0, 3, 1, 2).copy())


def mc_predict(model, tiles_u8, mc_n, seed, tile_index0=0):
    """N stochastic passes, ``reduce_mean`` / ``reduce_std`` (population) over them: (mean[n,2], std[n,2])."""
    feat = model.features(per_image_standardization(tiles_u8))
    idx = np.arange(tile_index0, tile_index0 + feat.shape[0])
    st = np.stack([model.head_pass(feat, idx, p, seed).numpy().astype(np.float64) for p in range(mc_n)])
    mean = st.mean(axis=0)
    std = np.sqrt(((st - mean) ** 2).mean(axis=0))
    return mean.astype(np.float32), std.astype(np.float32)
