"""Philox4x32-10 counter RNG + the dropout-mask contract (numpy, CPU oracle).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference leaves dropout active at inference (``biscuit/experiment.py:849``
sets ``hp.uq = True``; Slideflow then builds the head with dropout layers that
are hard-wired ``training=True``) and draws masks from TensorFlow's stateful RNG,
which cannot be reproduced without TensorFlow.  "Fixed dropout seed" therefore
means this build's own counter-based contract, bit-identical on CPU and GPU:

    r    = philox4x32_10(counter=(unit // 4, layer, mc_pass, tile_index),
                         key=(seed & 0xffffffff, seed >> 32))[unit % 4]
    keep = r >= floor(rate * 2**32)          (integer compare, P(keep) = 1 - rate)
    y    = x * keep * fp32(1 / (1 - rate))   (inverted dropout, hp.py:11 rate 0.1)

Philox4x32-10 follows Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"
(SC'11); pinned below against the Random123 known-answer vectors.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  All args broadcastable uint32-valued
    arrays (or ints); returns four uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *[np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def keep_threshold(rate):
    """Integer threshold t such that ``r >= t`` keeps a unit (P = 1 - rate)."""
    return int(np.floor(float(rate) * 4294967296.0))


def dropout_keep(seed, tile_index, mc_pass, layer, n_units, rate):
    """Boolean keep mask of shape ``tile_index.shape + (n_units,)``.

    tile_index: int array of *global* tile indices; mc_pass: int (or array
    broadcastable to tile_index); layer: 0, 1, 2 for the three head dropouts.
    """
    tile_index = np.asarray(tile_index, dtype=np.int64)
    mc_pass = np.broadcast_to(np.asarray(mc_pass, dtype=np.int64), tile_index.shape)
    assert n_units % 4 == 0
    groups = np.arange(n_units // 4, dtype=np.uint64)
    shp = tile_index.shape + (n_units // 4,)
    c0 = np.broadcast_to(groups, shp)
    c1 = np.full(shp, layer, dtype=np.uint64)
    c2 = np.broadcast_to(mc_pass[..., None].astype(np.uint64), shp)
    c3 = np.broadcast_to(tile_index[..., None].astype(np.uint64), shp)
    seed = int(seed)
    r = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=-1).reshape(tile_index.shape + (n_units,))
    return words >= np.uint32(keep_threshold(rate)) if keep_threshold(rate) < 2**32 \
        else np.zeros_like(words, dtype=bool)


def dropout_scale(rate):
    return np.float32(1.0 / (1.0 - float(rate)))
