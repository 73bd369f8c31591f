"""Weights for the MC-dropout Xception classifier: synthetic generation (seeded),
BN folding and packing into the device layout ``libbiscuit_hip.so`` consumes.

The reference never stores weights in its repository; a trained model is a
Slideflow/Keras SavedModel located by ``biscuit/utils.py:233-272`` (``find_model``).
The canonical in-memory form here is a dict of float32 numpy arrays keyed by Keras
layer/variable names (``block4_sepconv2/pointwise_kernel`` ...), so a converter from
a real SavedModel only has to fill the same dict.  Architecture and sizes follow
``biscuit/hp.py:3-23`` (xception, include_top=False, pooling='avg',
hidden_layers=2, hidden_layer_width=1024) and keras.applications.Xception.

Device layout ("BQW1" blob, build-owned, parsed by csrc/weights.cpp):
  * matrix-core layers (pointwise 1x1, residual 1x1/s2, block1_conv2 as im2col,
    hidden_0/hidden_1): weights pre-swizzled into MFMA operand-fragment order
        wp[nf][kb][lane][v] = W[k = kb*2V + (lane>>5)*V + v][n = nf*32 + (lane&31)]
    (V = 8 for bf16 / 32x32x16, V = 4 for fp32 / 4x 32x32x2), so one wave
    instruction loads one whole 1 KiB fragment, fully coalesced, with no LDS hop;
  * folded BatchNorm as per-output-channel fp32 ``scale``/``bias`` applied in the
    kernel epilogue (y = acc*s + b, s = gamma/sqrt(var+eps), b = beta - mean*s);
  * depthwise 3x3 kernels as fp32 [9][C_padded];
  * 728-channel tensors are padded to 736 (= 46*16) channels with zero weights.
"""
import struct

import numpy as np

BN_EPS = 1e-3
ENTRY = [(2, 64, [128, 128]), (3, 128, [256, 256]), (4, 256, [728, 728])]
MAGIC = b'BQW1'


def pad_channels(c):
    """Device channel stride of a c-channel activation (multiple of 16)."""
    return (c + 15) // 16 * 16


def sepconv_plan():
    """[(name, cin, cout)] of the 34 separable convolutions, in execution order."""
    plan = []
    for block, cin, chans in ENTRY:
        c = cin
        for i, co in enumerate(chans, 1):
            plan.append((f'block{block}_sepconv{i}', c, co))
            c = co
    for block in range(5, 13):
        for i in (1, 2, 3):
            plan.append((f'block{block}_sepconv{i}', 728, 728))
    plan += [('block13_sepconv1', 728, 728), ('block13_sepconv2', 728, 1024),
             ('block14_sepconv1', 1024, 1536), ('block14_sepconv2', 1536, 2048)]
    return plan


def residual_plan():
    return [('block2_res', 64, 128), ('block3_res', 128, 256),
            ('block4_res', 256, 728), ('block13_res', 728, 1024)]


def _bn(rng, w, name, c, gamma=1.0):
    w[name + '/gamma'] = (gamma * rng.uniform(0.9, 1.1, c)).astype(np.float32)
    w[name + '/beta'] = rng.normal(0.0, 0.05, c).astype(np.float32)
    w[name + '/moving_mean'] = rng.normal(0.0, 0.05, c).astype(np.float32)
    w[name + '/moving_variance'] = rng.uniform(0.9, 1.1, c).astype(np.float32)


def synthetic_weights(seed=1, n_classes=2, logit_gain=0.3, hard=False):
    """Seeded random-init weights of the hp.nature2022 architecture.

    ``hard=True`` is the parity stress set: BatchNorm moving variances log-uniform in [0.25, 4] with the
    convolution in front scaled so that its output really has that variance (as a trained network's
    statistics do), moving means and betas far from zero, gammas in [0.8, 1.25], and -- unless
    ``logit_gain`` is given explicitly -- an output layer with O(1) logits (gain 1.0).  A folding mistake
    (eps, sqrt, sign of the mean) or a lost bf16 bit moves these predictions visibly; the default set
    (variances and gammas within 10 % of one, logits within +-0.6) would hide it.

    Variance-preserving init so activations stay O(1) through the 36 conv layers with
    BatchNorm in inference mode (moving statistics ~ identity): conv / pointwise
    kernels ~ N(0, g/fan_in) with g = 2 behind a ReLU and 1 otherwise, depthwise
    kernels ~ N(0, 1/9); the last BN of every residual branch has a damped gamma so
    the residual stream grows slowly.  There is no network access for a checkpoint.
    """
    rng = np.random.default_rng(seed)
    w = {}

    def conv(name, kh, cin, cout, gain):
        std = np.sqrt(gain / (kh * kh * cin))
        w[name + '/kernel'] = rng.normal(0, std, (kh, kh, cin, cout)).astype(np.float32)

    def sep(name, cin, cout, gain):
        w[name + '/depthwise_kernel'] = rng.normal(0, 1 / 3.0, (3, 3, cin, 1)).astype(np.float32)
        w[name + '/pointwise_kernel'] = rng.normal(
            0, np.sqrt(gain / cin), (1, 1, cin, cout)).astype(np.float32)

    conv('block1_conv1', 3, 3, 32, 1.0)
    _bn(rng, w, 'block1_conv1_bn', 32)
    conv('block1_conv2', 3, 32, 64, 2.0)
    _bn(rng, w, 'block1_conv2_bn', 64)
    for name, cin, cout in residual_plan():
        conv(name + '_conv', 1, cin, cout, 1.0)
        _bn(rng, w, name + '_bn', cout, gamma=0.7)
    for name, cin, cout in sepconv_plan():
        block = int(name[5:name.index('_')])
        idx = int(name[-1])
        last = (idx == 3) or (block in (2, 3, 4, 13) and idx == 2)
        sep(name, cin, cout, 2.0)
        gamma = (0.3 if 5 <= block <= 12 else 0.7) if last else 1.0
        _bn(rng, w, name + '_bn', cout, gamma=gamma)
    # head: hidden_layers=2, hidden_layer_width=1024 (hp.py:13,21); GAP features are
    # post-ReLU means (non-negative, strongly correlated), so centre the first layer.
    k0 = rng.normal(0, np.sqrt(2.0 / 2048), (2048, 1024)).astype(np.float32)
    w['hidden_0/kernel'] = k0 - k0.mean(axis=0, keepdims=True)
    w['hidden_0/bias'] = rng.normal(0, 0.1, 1024).astype(np.float32)
    w['hidden_1/kernel'] = rng.normal(0, np.sqrt(2.0 / 1024), (1024, 1024)).astype(np.float32)
    w['hidden_1/bias'] = rng.normal(0, 0.1, 1024).astype(np.float32)
    k2 = rng.normal(0, np.sqrt(1.0 / 1024), (1024, n_classes)).astype(np.float32)
    w['logits/kernel'] = (k2 - k2.mean(axis=1, keepdims=True)) * np.float32(logit_gain)
    w['logits/bias'] = np.zeros(n_classes, np.float32)
    if hard:
        _harden(w, seed, n_classes, logit_gain)
    return w


def _harden(w, seed, n_classes, logit_gain):
    """In place: the ``hard=True`` statistics of ``synthetic_weights`` (own random stream, so the default
    set stays bit-identical)."""
    rng = np.random.default_rng([seed, 0x4841524])
    for name in sorted(k[:-len('/gamma')] for k in w if k.endswith('/gamma')):
        c = w[name + '/gamma'].size
        var = np.exp(rng.uniform(np.log(0.25), np.log(4.0), c)).astype(np.float32)
        sd = np.sqrt(var)
        conv = name[:-len('_bn')] if not name.endswith('_res_bn') else name[:-len('_bn')] + '_conv'
        key = conv + ('/pointwise_kernel' if conv + '/pointwise_kernel' in w else '/kernel')
        w[key] = (w[key] * sd).astype(np.float32)                       # output channel c now has std sd[c]
        w[name + '/moving_variance'] = var
        w[name + '/moving_mean'] = (rng.normal(0.0, 0.3, c) * sd).astype(np.float32)
        w[name + '/beta'] = rng.normal(0.0, 0.2, c).astype(np.float32)
        w[name + '/gamma'] = (w[name + '/gamma'] * rng.uniform(0.8, 1.25, c)).astype(np.float32)
    if logit_gain == 0.3:                                                # the default was not overridden
        w['logits/kernel'] = (w['logits/kernel'] * np.float32(1.0 / 0.3)).astype(np.float32)
    w['logits/bias'] = rng.normal(0.0, 0.2, n_classes).astype(np.float32)


def expected_shapes(n_classes=2):
    """{variable name: shape} of every tensor the hp.nature2022 classifier needs (Keras layout)."""
    sh = {'block1_conv1/kernel': (3, 3, 3, 32), 'block1_conv2/kernel': (3, 3, 32, 64)}

    def bn(name, c):
        for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            sh[f'{name}/{v}'] = (c,)
    bn('block1_conv1_bn', 32)
    bn('block1_conv2_bn', 64)
    for name, cin, cout in residual_plan():
        sh[name + '_conv/kernel'] = (1, 1, cin, cout)
        bn(name + '_bn', cout)
    for name, cin, cout in sepconv_plan():
        sh[name + '/depthwise_kernel'] = (3, 3, cin, 1)
        sh[name + '/pointwise_kernel'] = (1, 1, cin, cout)
        bn(name + '_bn', cout)
    sh.update({'hidden_0/kernel': (2048, 1024), 'hidden_0/bias': (1024,), 'hidden_1/kernel': (1024, 1024),
               'hidden_1/bias': (1024,), 'logits/kernel': (1024, n_classes), 'logits/bias': (n_classes,)})
    return sh


def validate(w, n_classes=2):
    """Raise ValueError unless ``w`` holds exactly the expected tensors with the expected shapes."""
    want = expected_shapes(n_classes)
    missing = sorted(set(want) - set(w))
    extra = sorted(set(w) - set(want))
    bad = sorted(k for k in want if k in w and tuple(np.shape(w[k])) != want[k])
    if missing or extra or bad:
        raise ValueError(f'weights do not match the hp.nature2022 classifier: missing {missing[:4]}, '
                         f'unexpected {extra[:4]}, wrong shape {bad[:4]}')
    if not all(np.isfinite(np.asarray(w[k], dtype=np.float32)).all() for k in want):
        raise ValueError('weights contain NaN/Inf')
    return w


def save_npz(path, w):
    """Portable weight file: one float32 array per Keras variable name.  A converter from a
    Slideflow/Keras SavedModel only has to write this (it needs TensorFlow, which is not available
    in the build environment)."""
    np.savez_compressed(path, **{k.replace('/', '__'): np.asarray(v, np.float32) for k, v in validate(w).items()})


def load_npz(path):
    with np.load(path) as z:
        return validate({k.replace('__', '/'): z[k].astype(np.float32) for k in z.files})


def count_backbone_params(w):
    head = ('hidden_0', 'hidden_1', 'logits')
    return int(sum(v.size for k, v in w.items() if not k.startswith(head)))


# ---------------------------------------------------------------------------
# device packing
# ---------------------------------------------------------------------------
def f32_to_bf16_bits(a):
    """Round-to-nearest-even float32 -> bfloat16 bit patterns (uint16)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)
    return r.astype(np.uint16)


def f32_to_f16_bits(a):
    """Round-to-nearest-even float32 -> IEEE half bit patterns (uint16); values beyond +-65504 saturate, as the
    device's conversions do (MODE.FP16_OVFL)."""
    a = np.clip(np.ascontiguousarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).view(np.uint16)


HEAD_SPLIT_SCALE = 2048.0          # 2^11: kernels_head.hip HSCALE


def split_f16(a):
    """float32 -> (hi, lo) IEEE-half bit patterns with a ~= hi + lo / 2^11: hi = f16(a), lo = f16((a - hi) * 2^11)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    hi = np.clip(a, -65504.0, 65504.0).astype(np.float16)
    lo = ((a - hi.astype(np.float32)) * np.float32(HEAD_SPLIT_SCALE)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


DTYPE_CODE = {'f32': 0, 'bf16': 1, 'f16': 2}      # BQ_DTYPE_* of include/biscuit_hip.h


def nfrags_padded(n):
    nf = (n + 31) // 32
    return nf if nf <= 8 else (nf + 7) // 8 * 8


def pack_fragments(wkn, kpad, vec):
    """W[K][N] -> fragment order [NFp][KB][64][vec] float32 (zero padded)."""
    k, n = wkn.shape
    assert kpad % (2 * vec) == 0 and kpad >= k
    nfp = nfrags_padded(n)
    full = np.zeros((kpad, nfp * 32), np.float32)
    full[:k, :n] = wkn
    kb = kpad // (2 * vec)
    # k = kb*2V + h*V + v ; n = nf*32 + r ; lane = h*32 + r
    t = full.reshape(kb, 2, vec, nfp, 32)          # [kb][h][v][nf][r]
    t = t.transpose(3, 0, 1, 4, 2)                 # [nf][kb][h][r][v]
    return np.ascontiguousarray(t).reshape(nfp, kb, 64, vec)


def frag16_channel(nf, r):
    """Output channel that row r of 16-wide n-fragment nf computes in kernels_wide.hip.  The two fragments 2q, 2q+1 of a
    pair interleave in groups of four channels: the accumulator layout of v_mfma_f32_16x16x32 gives a lane (pixel l & 15,
    group g = l >> 4) rows 4g .. 4g+3 of a fragment, so with this order it holds, over the pair, the EIGHT consecutive
    channels 32q + 8g .. 32q + 8g + 7 of its pixel -- one 16-byte store per pair straight from the accumulators."""
    return 32 * (nf // 2) + 8 * (r // 4) + 4 * (nf % 2) + (r % 4)


def pack_fragments16(wkn, kpad, npad):
    """W[K][N] -> v_mfma_f32_16x16x32 fragment order [KS][NF16][64][8] float32 (zero padded): k-step ks, 16-wide
    n-fragment nf, lane = kg*16 + r holds W[k = ks*32 + kg*8 + v][n = frag16_channel(nf, r)] in element v.  All fragments of
    one k-step are contiguous (48 KiB for 768 outputs), so a wave's 6 fragments are one 6 KiB run."""
    k, n = wkn.shape
    assert kpad % 32 == 0 and kpad >= k and npad % 32 == 0 and npad >= n
    full = np.zeros((kpad, npad), np.float32)
    full[:k, :n] = wkn
    nf, r = np.meshgrid(np.arange(npad // 16), np.arange(16), indexing='ij')
    full = full[:, frag16_channel(nf, r).reshape(-1)]       # column nf*16 + r <- channel frag16_channel(nf, r)
    t = full.reshape(kpad // 32, 4, 8, npad // 16, 16)      # [ks][kg][v][nf][r]
    t = t.transpose(0, 3, 1, 4, 2)                          # [ks][nf][kg][r][v]
    return np.ascontiguousarray(t).reshape(kpad // 32, npad // 16, 64, 8)


def pack_fragments32(wkn, kpad, npad):
    """W[K][N] -> v_mfma_f32_32x32x16 fragment order [KS][NF32][64][8] float32 (zero padded): k-step ks (16 deep), 32-wide
    n-fragment nf, lane = h*32 + r holds W[k = ks*16 + h*8 + v][n = nf*32 + r] in element v.  All fragments of one
    k-step are contiguous."""
    k, n = wkn.shape
    assert kpad % 16 == 0 and kpad >= k and npad % 32 == 0 and npad >= n
    full = np.zeros((kpad, npad), np.float32)
    full[:k, :n] = wkn
    t = full.reshape(kpad // 16, 2, 8, npad // 32, 32)      # [ks][h][v][nf][r]
    t = t.transpose(0, 3, 1, 4, 2)                          # [ks][nf][h][r][v]
    return np.ascontiguousarray(t).reshape(kpad // 16, npad // 32, 64, 8)


def wide_layers():
    """The separable convolutions of kernels_wide.hip (8 waves, 16x16x32 fragment order with interleaved n-fragment pairs):
    the 26 layers 728 -> 728 -- block4_sepconv2 (37x37 maps), blocks 5-12 and block13_sepconv1 (19x19 maps) --,
    block4_sepconv1 (256 -> 728, 37x37), block3_sepconv1 / 2 (128 / 256 -> 256, 74x74) and block13_sepconv2 (728 -> 1024,
    19x19, as two launches of 512 columns)."""
    return (['block3_sepconv1', 'block3_sepconv2', 'block4_sepconv1', 'block4_sepconv2'] +
            [f'block{b}_sepconv{i}' for b in range(5, 13) for i in (1, 2, 3)] + ['block13_sepconv1', 'block13_sepconv2'])


# kernels_stream.hip (round 4): the 147x147 separable convolutions of block 2 take the same 16x16x32 fragment order
STREAM_LAYERS = ('block2_sepconv1', 'block2_sepconv2')      # (block3_sepconv1 has its wp16 as a wide layer)
TAIL_RES_LAYERS = ('block2_res', 'block3_res')
# kernels_exit.hip (round 4): block 14's pointwise GEMMs, the same fragment order
EXIT_LAYERS = ('block14_sepconv1', 'block14_sepconv2')


# ---------------------------------------------------------------------------
# activation exponents: keeping 16-bit storage in range by construction
# ---------------------------------------------------------------------------
# Between its weights the network is positively homogeneous: convolutions are linear, ReLU and max-pooling commute with a
# positive factor, and a folded BatchNorm y = s * conv(x) + b turns "input stored as x / 2^kin, output wanted as y / 2^kout" into
# s' = s * 2^(kin - kout), b' = b / 2^kout -- exact in fp32, no weight matrix touched, every mantissa the kernels produce
# unchanged (a power of two moves exponents only).  So every STORED tensor can carry its own power-of-two exponent, chosen so
# that it peaks well inside IEEE half's range, as long as tensors that meet in an addition share one: the two branches of a
# block with a strided shortcut, and the whole residual stream of the middle flow (blocks 4 .. 12: the identity shortcut is
# added as stored).  The pooled features come back to true scale in the pooling epilogue ("act/feat_mul" = 2^k).
def tensor_plan():
    """[(layer, input tensor, output tensor)] over the stored tensors of the backbone; tensors that share an exponent share a
    name (``block{2,3,4,13}_out``: main branch, shortcut and sum; ``block4_out``: the residual stream through block 12)."""
    plan = [('block1_conv1', None, 'block1_conv1'), ('block1_conv2', 'block1_conv1', 'block1_conv2')]
    x = 'block1_conv2'
    for b in (2, 3, 4):
        plan += [(f'block{b}_sepconv1', x, f'block{b}_sepconv1'), (f'block{b}_sepconv2', f'block{b}_sepconv1', f'block{b}_out'),
                 (f'block{b}_res', x, f'block{b}_out')]
        x = f'block{b}_out'
    for b in range(5, 13):
        plan += [(f'block{b}_sepconv1', x, f'block{b}_sepconv1'), (f'block{b}_sepconv2', f'block{b}_sepconv1', f'block{b}_sepconv2'),
                 (f'block{b}_sepconv3', f'block{b}_sepconv2', x)]
    plan += [('block13_sepconv1', x, 'block13_sepconv1'), ('block13_sepconv2', 'block13_sepconv1', 'block13_out'),
             ('block13_res', x, 'block13_out'),
             ('block14_sepconv1', 'block13_out', 'block14_sepconv1'), ('block14_sepconv2', 'block14_sepconv1', 'block14_sepconv2')]
    return plan


FEATURE_TENSOR = 'block14_sepconv2'      # what the global average pool reads


def tensor_taps():
    """{exponent-sharing tensor name: [debug-tap names whose stored values carry that exponent]} (bq_debug_activation)."""
    taps = {}
    for layer, _, out in tensor_plan():
        taps.setdefault(out, []).append(layer)
    for b in (2, 3, 4, 13):
        taps[f'block{b}_out'].append(f'block{b}_out')
    taps['block4_out'] += [f'block{b}_out' for b in range(5, 13)]
    # (a sepconv3 output only exists as the sum with the stream: its tap name is the block's output)
    taps['block4_out'] = [t for t in taps['block4_out'] if not t.endswith('_sepconv3')]
    return taps


def depthwise_gain(w, layer):
    """max over channels of the l1 norm of ``layer``'s 3x3 depthwise taps: |dw(x)| <= gain * max|x|, the bound on the
    depthwise result every fused kernel rounds to the storage type before the pointwise product."""
    k = w.get(layer + '/depthwise_kernel')
    return 1.0 if k is None else float(np.abs(np.asarray(k, np.float64)).reshape(9, -1).sum(0).max())


def choose_act_exponents(w, peaks, target_log2=12):
    """Power-of-two exponents {tensor: k >= 0} from measured peaks {tensor: max |value|, true scale} (``Engine.calibrate``):
    the smallest k with peak / 2^k <= 2^target_log2, where a tensor's peak includes the depthwise results of the separable
    convolutions that read it (``depthwise_gain``).  2^12 leaves a factor 16 to IEEE half's 65504 for tiles unlike the
    calibration batch.  Tensors never scale UP: a network that fits as it is keeps exponent 0 everywhere and packs to the
    same blob as without exponents."""
    need = {}
    for layer, tin, tout in tensor_plan():
        need[tout] = max(need.get(tout, 0.0), float(peaks.get(tout, 0.0)))
        if tin is not None:
            need[tin] = max(need.get(tin, 0.0), float(peaks.get(tin, 0.0)) * depthwise_gain(w, layer))
    lim = 2.0 ** target_log2
    return {t: (0 if p <= lim else int(np.ceil(np.log2(p / lim)))) for t, p in need.items()}


def equivalent_rescaled(w, factor):
    """The same classifier with every stored backbone tensor but the last ``factor`` times larger (any positive real), the way
    a trained network comes to large activations: the weights keep their size, the BatchNorm statistics grow.  A layer whose
    input tensor carries the factor gets moving_mean * f and moving_variance -> f^2 (var + eps) - eps (so that sqrt(var' + eps)
    = f sqrt(var + eps)); a layer whose output tensor carries it gets gamma * f and beta * f; the tensor the global pool reads
    stays at true scale, so the head is untouched.  In real arithmetic the function is unchanged; in IEEE-half storage the
    activations now overflow -- the test vehicle for the activation exponents, and a statement of the homogeneity they rest on."""
    f = np.float64(factor)
    out = {k: np.array(v, np.float32, copy=True) for k, v in w.items()}
    for layer, tin, tout in tensor_plan():
        bn = layer + '_bn'
        if tin is not None:             # (every tensor a layer reads is scaled: only the network's input is not)
            out[bn + '/moving_mean'] = (out[bn + '/moving_mean'].astype(np.float64) * f).astype(np.float32)
            var = out[bn + '/moving_variance'].astype(np.float64)
            out[bn + '/moving_variance'] = (f * f * (var + BN_EPS) - BN_EPS).astype(np.float32)
        if tout != FEATURE_TENSOR:
            out[bn + '/gamma'] = (out[bn + '/gamma'].astype(np.float64) * f).astype(np.float32)
            out[bn + '/beta'] = (out[bn + '/beta'].astype(np.float64) * f).astype(np.float32)
    return out


def fold_bn(w, name):
    s = w[name + '/gamma'] / np.sqrt(w[name + '/moving_variance'] + np.float32(BN_EPS))
    b = w[name + '/beta'] - w[name + '/moving_mean'] * s
    return s.astype(np.float32), b.astype(np.float32)


def _padvec(v, n):
    out = np.zeros(n, np.float32)
    out[:v.size] = v
    return out


def pack_blob(w, dtype='bf16', act_exp=None):
    """Fold BN and serialise every tensor the device needs into one BQW1 blob.  ``act_exp``: {tensor: k} activation
    exponents (``choose_act_exponents``): the stored tensor is the network's divided by 2^k; missing tensors and None: 0."""
    assert dtype in DTYPE_CODE
    act_exp = {t: int(k) for t, k in (act_exp or {}).items() if int(k) != 0}
    io = {layer: (tin, tout) for layer, tin, tout in tensor_plan()}
    unknown = set(act_exp) - {t for _, _, t in tensor_plan()}
    if unknown:
        raise ValueError(f'activation exponents for unknown tensors: {sorted(unknown)}')

    def fold(layer):
        """folded BN of ``layer`` with the exponents of the tensors it reads and writes applied (exact: powers of two)"""
        s, b = fold_bn(w, layer + '_bn')
        tin, tout = io[layer]
        kin, kout = act_exp.get(tin, 0), act_exp.get(tout, 0)
        if kin or kout:
            s = np.ldexp(s, kin - kout).astype(np.float32)
            b = np.ldexp(b, -kout).astype(np.float32)
        return s, b
    half = dtype != 'f32'                       # a 16-bit matrix-core type
    vec = 8 if half else 4
    to_bits = {'bf16': f32_to_bf16_bits, 'f16': f32_to_f16_bits}.get(dtype)
    entries = []

    def add(name, arr):
        entries.append((name, np.ascontiguousarray(arr)))

    def add_mat(name, wkn, kpad):
        p = pack_fragments(wkn, kpad, vec)
        add(name + '/wp', to_bits(p) if half else p)
        return p.shape[0] * 32

    def add_affine(name, s, b, npad):
        add(name + '/scale', _padvec(s, npad))
        add(name + '/bias', _padvec(b, npad))

    # stem conv1 on the vector ALU: fp32 [27][32], k = (dy*3+dx)*3 + c
    add('block1_conv1/w', w['block1_conv1/kernel'].reshape(27, 32))
    s, b = fold('block1_conv1')
    add_affine('block1_conv1', s, b, 32)
    if half:
        # kernels_front.hip (round 4): the same weights for the matrix cores -- 16x16x32 fragment order, K = 27 padded to 32,
        # every fp32 weight as two IEEE halves hi + lo / 2^11 (f16 MFMAs for both 16-bit storage types): [hi, lo][2][64][8]
        hi, lo = split_f16(pack_fragments16(w['block1_conv1/kernel'].reshape(27, 32), 32, 32))
        add('block1_conv1/w16', np.concatenate([hi.reshape(-1), lo.reshape(-1)]))
    # stem conv2 as im2col GEMM: k = (dy*3+dx)*32 + c
    npad = add_mat('block1_conv2', w['block1_conv2/kernel'].reshape(288, 64), 288)
    if half:   # ... and in 16x16x32 fragment order, one k-step per tap (kernels_front.hip)
        add('block1_conv2/wp16', to_bits(pack_fragments16(w['block1_conv2/kernel'].reshape(288, 64), 288, 64)))
    s, b = fold('block1_conv2')
    add_affine('block1_conv2', s, b, npad)
    for name, cin, cout in residual_plan():
        npad = add_mat(name, w[name + '_conv/kernel'].reshape(cin, cout), pad_channels(cin))
        if half and npad % 128 == 0 and cin <= 128:   # kernels_respool.hip: shortcut conv + max-pool + add
            # in one kernel (blocks 2 and 3; the wider shortcuts measured faster as two kernels)
            add(name + '/wp32', to_bits(pack_fragments32(w[name + '_conv/kernel'].reshape(cin, cout),
                                                                  pad_channels(cin), npad)))
        if half and name in TAIL_RES_LAYERS:          # kernels_stream.hip: the shortcut inside the fused block tail
            add(name + '/wp16', to_bits(pack_fragments16(w[name + '_conv/kernel'].reshape(cin, cout), pad_channels(cin), npad)))
        s, b = fold(name)
        add_affine(name, s, b, npad)
    wide = set(wide_layers())
    for name, cin, cout in sepconv_plan():
        cp = pad_channels(cin)
        dw = np.zeros((9, cp), np.float32)
        dw[:, :cin] = w[name + '/depthwise_kernel'].reshape(9, cin)
        add(name + '/dw', dw)
        npad = add_mat(name, w[name + '/pointwise_kernel'].reshape(cin, cout), cp)
        if half and (name in wide or name in STREAM_LAYERS or name in EXIT_LAYERS) and cp % 32 == 0:
            add(name + '/wp16', to_bits(pack_fragments16(w[name + '/pointwise_kernel'].reshape(cin, cout), cp, npad)))
        s, b = fold(name)
        add_affine(name, s, b, npad)
    if act_exp.get(FEATURE_TENSOR, 0):
        # the pooled features return to true scale in the pooling epilogue: means * 2^k (c->feat_mul)
        add('act/feat_mul', np.array([2.0 ** act_exp[FEATURE_TENSOR]], np.float32))
    # The head keeps fp32 ACCURACY regardless of the backbone dtype (MC std ~1e-2 must not be quantisation noise), at the
    # 16-bit matrix rate: every weight is split into two IEEE halves, w = hi + lo / 2^11 (22 significand bits; the scale
    # keeps lo out of the subnormals), both in 32x32x16 fragment order (kernels_head.hip multiplies three of the four
    # hi / lo cross terms of x * w with fp32 accumulation).
    for name, kin in (('hidden_0', 2048), ('hidden_1', 1024)):
        p = pack_fragments(w[name + '/kernel'], kin, 8)
        hi, lo = split_f16(p)
        add(name + '/wph', hi)
        add(name + '/wpl', lo)
        add(name + '/bias', _padvec(w[name + '/bias'].astype(np.float32), p.shape[0] * 32))
    add('logits/w', w['logits/kernel'].astype(np.float32))
    add('logits/bias', w['logits/bias'].astype(np.float32))

    # header + directory + 256-B aligned payloads
    hdr_size = 16 + 64 * len(entries)
    off = (hdr_size + 255) // 256 * 256
    directory, payload = [], []
    for name, arr in entries:
        raw = arr.tobytes()
        nm = name.encode()
        assert len(nm) < 48
        directory.append(struct.pack('<48sQQ', nm, off, len(raw)))
        pad = (-len(raw)) % 256
        payload.append(raw + b'\0' * pad)
        off += len(raw) + pad
    head = struct.pack('<4sIII', MAGIC, 1, len(entries), DTYPE_CODE[dtype])
    blob = head + b''.join(directory)
    blob += b'\0' * ((-len(blob)) % 256)
    return blob + b''.join(payload)
