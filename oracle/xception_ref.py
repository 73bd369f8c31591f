"""fp32 CPU restatement of the tile-level MC-dropout inference path (PyTorch-CPU
functional ops + numpy).  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

PARITY UNPINNED on this (producer) side: the arithmetic lives in Slideflow /
Keras, which the reference neither vendors nor pins (``requirements.txt:1,5``)
and which cannot be installed here.  What is restated, and from where:

* network + preprocessing spec ........ ``biscuit/hp.py:3-23``  (xception, 299 px,
  ``pooling='avg'``, ``include_top=False``, 2 hidden Dense(1024), dropout 0.1)
* preprocess order and output tuple ... ``results.py:250-258``  (normalise ->
  ``tf.image.per_image_standardization`` -> model -> ``(mean, std)``)
* which column is y_pred / uncertainty  ``biscuit/utils.py:19-28`` (class-1 mean,
  class-1 std)
* Keras ``applications.Xception`` layer graph, ``StaticDropout`` (always on),
  ``get_uq_predictions`` loop (N forward passes, ``reduce_mean`` / ``reduce_std``)
  -- published Keras / Slideflow 1.1 algorithm, restated.

Weights arrive as a dict of numpy arrays in Keras layout/names
(``blockB_sepconvI/depthwise_kernel`` [3,3,C,1], ``.../pointwise_kernel``
[1,1,Cin,Cout], ``*_bn/{gamma,beta,moving_mean,moving_variance}``, conv
``kernel`` HWIO, dense ``kernel`` [in,out] + ``bias``).

``emulate='f16'`` / ``emulate='bf16'`` (or the older ``emulate_bf16=True``) additionally round to
that 16-bit type at exactly the points where the HIP path of that storage type does (staged
tile, every layer output, the depthwise result, the matrix-core weights; f16 saturates at
+-65504 as the device does); accumulation, depthwise taps and folded BN stay fp32.  These are
secondary checkers for the 16-bit kernels; the fp32 mode is the parity oracle.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox

BN_EPS = 1e-3          # keras.layers.BatchNormalization default epsilon
TILE_PX = 299          # hp.py:5
N_FEATURES = 2048

# (block, n_sepconv, channel plan) of keras.applications.Xception
ENTRY = [(2, [128, 128]), (3, [256, 256]), (4, [728, 728])]
MIDDLE = list(range(5, 13))
EXIT13 = [728, 1024]
EXIT14 = [1536, 2048]


F16_MAX = 65504.0


def _q(x, on):
    """Round to the storage type of the 16-bit HIP paths (RNE) and back.

    ``on`` is False/None (fp32), True/'bf16' (bfloat16) or 'f16' (IEEE half; the device
    runs with MODE.FP16_OVFL set, so an overflow saturates at +-65504 instead of inf).
    """
    if not on:
        return x
    if on == 'f16':
        return x.clamp(-F16_MAX, F16_MAX).to(torch.float16).to(torch.float32)
    return x.to(torch.bfloat16).to(torch.float32)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def standardize(tiles_u8):
    """tf.image.per_image_standardization on uint8 NHWC tiles (results.py:256).

    (x - mean) / max(std, 1/sqrt(N)), statistics over all H*W*C values of one
    image, population std.  Returns float32 NCHW.
    """
    x = torch.from_numpy(np.ascontiguousarray(tiles_u8)).to(torch.float32)
    n = x.shape[0]
    flat = x.reshape(n, -1)
    num = flat.shape[1]
    mean = flat.double().mean(dim=1)
    var = (flat.double() - mean[:, None]).pow(2).mean(dim=1)
    std = var.sqrt()
    adj = torch.maximum(std, torch.tensor(1.0 / np.sqrt(num), dtype=torch.float64))
    out = (x - mean.float()[:, None, None, None]) * (1.0 / adj).float()[:, None, None, None]
    return out.permute(0, 3, 1, 2).contiguous()


class XceptionOracle:
    def __init__(self, weights, dropout=0.1, emulate_bf16=False, threads=None, emulate=None):
        self.w = weights
        self.rate = float(dropout)
        # storage type emulated: None (fp32, THE parity oracle), 'bf16' or 'f16'
        if emulate not in (None, 'bf16', 'f16'):
            raise ValueError(f'emulate must be None, "bf16" or "f16", not {emulate!r}')
        self.bf = emulate if emulate else ('bf16' if emulate_bf16 else None)
        if threads:
            torch.set_num_threads(int(threads))

    # -- building blocks ---------------------------------------------------
    def _bn(self, x, name):
        w = self.w
        g, b = _t(w[name + '/gamma']), _t(w[name + '/beta'])
        m, v = _t(w[name + '/moving_mean']), _t(w[name + '/moving_variance'])
        # same folded form the device uses: y = x*s + (b - m*s)
        s = g / torch.sqrt(v + BN_EPS)
        o = b - m * s
        return x * s[None, :, None, None] + o[None, :, None, None]

    def _conv(self, x, name, stride, quant_w):
        k = _t(self.w[name + '/kernel']).permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW
        return F.conv2d(x, _q(k, quant_w), stride=stride, padding=0)

    def _sepconv(self, x, name):
        dk = _t(self.w[name + '/depthwise_kernel'])            # [3,3,C,1]
        c = dk.shape[2]
        dk = dk.permute(2, 3, 0, 1).contiguous()               # [C,1,3,3]
        pk = _t(self.w[name + '/pointwise_kernel']).permute(3, 2, 0, 1).contiguous()
        d = F.conv2d(x, dk, padding=1, groups=c)               # 'same', zero pad
        d = _q(d, self.bf)
        return F.conv2d(d, _q(pk, self.bf))

    @staticmethod
    def _maxpool_same(x):
        """3x3 / stride 2 / 'same' max-pool with TensorFlow's asymmetric padding."""
        h, w = x.shape[2], x.shape[3]

        def pads(n):
            out = -(-n // 2)
            tot = max((out - 1) * 2 + 3 - n, 0)
            return tot // 2, tot - tot // 2
        (pt, pb), (pl, pr) = pads(h), pads(w)
        x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
        return F.max_pool2d(x, 3, 2)

    def _res(self, x, block):
        r = self._conv(x, f'block{block}_res_conv', 2, self.bf)   # 1x1 / s2 / 'same' == even indices
        return _q(self._bn(r, f'block{block}_res_bn'), self.bf)

    # -- backbone ------------------------------------------------------------
    def backbone(self, x, taps=None):
        """x: float32 NCHW standardised tiles -> [n, 2048] GAP features.

        ``taps`` (dict) receives named intermediate activations (NCHW) when given.
        """
        bf = self.bf

        def tap(name, t):
            if taps is not None:
                taps[name] = t.clone()
            return t

        x = _q(x, bf)
        tap('staged', x)
        # block 1 (stem).  conv1 runs on the vector ALU on the device with fp32
        # weights, so its weights are not rounded in bf16 emulation.
        x = F.relu(self._bn(self._conv(x, 'block1_conv1', 2, False), 'block1_conv1_bn'))
        x = tap('block1_conv1', _q(x, bf))
        x = F.relu(self._bn(self._conv(x, 'block1_conv2', 1, bf), 'block1_conv2_bn'))
        x = tap('block1_conv2', _q(x, bf))
        # entry flow blocks 2-4
        for block, chans in ENTRY:
            res = tap(f'block{block}_res', self._res(x, block))
            y = x if block == 2 else F.relu(x)
            y = self._bn(self._sepconv(y, f'block{block}_sepconv1'), f'block{block}_sepconv1_bn')
            y = tap(f'block{block}_sepconv1', _q(F.relu(y), bf))      # relu = next layer's *_act
            y = self._bn(self._sepconv(y, f'block{block}_sepconv2'), f'block{block}_sepconv2_bn')
            y = tap(f'block{block}_sepconv2', _q(y, bf))
            x = tap(f'block{block}_out', _q(self._maxpool_same(y) + res, bf))
        # middle flow blocks 5-12
        for block in MIDDLE:
            y = F.relu(x)
            for i in (1, 2, 3):
                y = self._bn(self._sepconv(y, f'block{block}_sepconv{i}'),
                             f'block{block}_sepconv{i}_bn')
                if i < 3:
                    y = _q(F.relu(y), bf)
            x = tap(f'block{block}_out', _q(y + x, bf))
        # exit flow
        res = self._res(x, 13)
        y = F.relu(x)
        y = self._bn(self._sepconv(y, 'block13_sepconv1'), 'block13_sepconv1_bn')
        y = _q(F.relu(y), bf)
        y = self._bn(self._sepconv(y, 'block13_sepconv2'), 'block13_sepconv2_bn')
        y = _q(y, bf)
        x = tap('block13_out', _q(self._maxpool_same(y) + res, bf))
        x = self._bn(self._sepconv(x, 'block14_sepconv1'), 'block14_sepconv1_bn')
        x = tap('block14_sepconv1', _q(F.relu(x), bf))
        x = self._bn(self._sepconv(x, 'block14_sepconv2'), 'block14_sepconv2_bn')
        x = tap('block14_sepconv2', _q(F.relu(x), bf))
        return x.mean(dim=(2, 3))                                  # pooling='avg' (hp.py:22)

    # -- head ----------------------------------------------------------------
    def head_pass(self, feat, tile_index, mc_pass, seed):
        """One stochastic pass of the head: dropout -> Dense(1024,relu) -> dropout ->
        Dense(1024,relu) -> dropout -> Dense(2) -> softmax.  Returns [n,2] float32."""
        w = self.w
        scale = float(philox.dropout_scale(self.rate))
        h = feat
        dims = [(N_FEATURES, 'hidden_0'), (1024, 'hidden_1'), (1024, 'logits')]
        for layer, (n_in, name) in enumerate(dims):
            keep = philox.dropout_keep(seed, tile_index, mc_pass, layer, n_in, self.rate)
            h = h * torch.from_numpy(keep.astype(np.float32)) * np.float32(scale)
            h = h @ _t(w[name + '/kernel']) + _t(w[name + '/bias'])
            if layer < 2:
                h = F.relu(h)
        return torch.softmax(h, dim=1)

    def mc_predict(self, tiles_u8, mc_n, seed, tile_index0=0, mode='head', batch=16):
        """The reference's get_uq_predictions loop.

        mode='full': ``for _ in range(mc_n): yp = model(img)`` -- the whole network is
        re-run every pass, exactly the reference's loop structure.
        mode='head': backbone once, head mc_n times.  Identical results because the
        only stochastic layers sit behind the global pool and BN is in inference mode.
        Returns (mean[n,2], std[n,2]) float32 numpy (population std, tf.math.reduce_std).
        """
        n = tiles_u8.shape[0]
        means, stds = [], []
        for s in range(0, n, batch):
            tb = tiles_u8[s:s + batch]
            idx = np.arange(tile_index0 + s, tile_index0 + s + tb.shape[0])
            x = standardize(tb)
            passes = []
            feat = self.backbone(x) if mode == 'head' else None
            for p in range(mc_n):
                f = feat if mode == 'head' else self.backbone(x)
                passes.append(self.head_pass(f, idx, p, seed))
            st = torch.stack(passes, dim=0)
            means.append(st.mean(dim=0).numpy())
            stds.append(st.std(dim=0, unbiased=False).numpy())
        return np.concatenate(means), np.concatenate(stds)

    def mc_from_features(self, feat, mc_n, seed, tile_index0=0):
        feat = torch.as_tensor(feat, dtype=torch.float32)
        idx = np.arange(tile_index0, tile_index0 + feat.shape[0])
        st = torch.stack([self.head_pass(feat, idx, p, seed) for p in range(mc_n)], 0)
        return st.mean(0).numpy(), st.std(0, unbiased=False).numpy()


def count_backbone_params(weights):
    """Keras counts conv/pointwise/depthwise kernels + 4 BN vectors per BN layer."""
    head = ('hidden_0', 'hidden_1', 'logits')
    return int(sum(v.size for k, v in weights.items() if not k.startswith(head)))
