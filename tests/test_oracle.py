"""CPU tests of the oracle itself: known-answer vectors, self-consistency checks that
substitute for the missing producer-side pins, small sizes only."""
import numpy as np
import pytest
import torch

from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights
from oracle import philox
from oracle.xception_ref import XceptionOracle, count_backbone_params, standardize


@pytest.fixture(scope='module')
def weights():
    return synthetic_weights(1)


@pytest.fixture(scope='module')
def tiles():
    return make_tiles(2, seed=5)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in got) == want


def test_dropout_mask_contract():
    idx = np.arange(3, 7)
    k = philox.dropout_keep(1234, idx, 2, 0, 2048, 0.1)
    assert k.shape == (4, 2048) and k.dtype == bool
    assert abs(k.mean() - 0.9) < 0.02
    # counter-based: independent of how tiles are batched
    k2 = philox.dropout_keep(1234, np.array([5]), 2, 0, 2048, 0.1)
    assert (k[2] == k2[0]).all()
    assert not (philox.dropout_keep(1235, idx, 2, 0, 2048, 0.1) == k).all()
    assert philox.dropout_keep(1, idx, 0, 0, 8, 0.0).all()
    assert philox.keep_threshold(0.1) == 429496729


def test_param_count(weights):
    # keras.applications.Xception(include_top=False) has 20,861,480 parameters
    assert count_backbone_params(weights) == 20861480
    head = sum(weights[k].size for k in weights if k.split('/')[0] in ('hidden_0', 'hidden_1', 'logits'))
    assert head == 2048 * 1024 + 1024 + 1024 * 1024 + 1024 + 1024 * 2 + 2


def test_standardize_matches_definition(tiles):
    x = standardize(tiles)
    assert x.shape == (2, 3, 299, 299)
    flat = tiles[0].astype(np.float64)
    want = (flat - flat.mean()) / max(flat.std(), 1 / np.sqrt(flat.size))
    np.testing.assert_allclose(x[0].permute(1, 2, 0).numpy(), want, atol=2e-6)
    const = np.full((1, 299, 299, 3), 7, np.uint8)       # std = 0 -> floor 1/sqrt(N)
    assert torch.all(standardize(const) == 0)


def test_maxpool_same_asymmetric_padding():
    # 74 -> 37 pads only at the end (TensorFlow 'same'), odd sizes pad (1,1)
    x = torch.arange(74 * 74, dtype=torch.float32).reshape(1, 1, 74, 74)
    y = XceptionOracle._maxpool_same(x)
    assert y.shape[-1] == 37
    assert y[0, 0, 0, 0] == x[0, 0, :3, :3].max()         # window starts at 0 (no leading pad)
    assert y[0, 0, 36, 36] == x[0, 0, 72:, 72:].max()
    z = XceptionOracle._maxpool_same(torch.arange(19 * 19, dtype=torch.float32).reshape(1, 1, 19, 19))
    assert z.shape[-1] == 10 and z[0, 0, 0, 0] == 20.0    # window rows/cols -1..1 -> max at (1,1)


def test_mc_semantics(weights, tiles):
    orc = XceptionOracle(weights)
    m_head, s_head = orc.mc_predict(tiles, 3, 1234, mode='head')
    m_full, s_full = orc.mc_predict(tiles, 3, 1234, mode='full')
    np.testing.assert_allclose(m_head, m_full, atol=1e-6)           # full == head
    np.testing.assert_allclose(s_head, s_full, atol=1e-6)
    np.testing.assert_allclose(m_head.sum(1), 1.0, atol=1e-6)       # softmax rows
    np.testing.assert_allclose(s_head[:, 0], s_head[:, 1], atol=1e-6)   # p0 + p1 = 1
    # batching / offset independence of the mask contract
    m_b, s_b = orc.mc_predict(tiles[1:], 3, 1234, tile_index0=1, mode='head')
    np.testing.assert_allclose(m_b[0], m_head[1], atol=1e-6)
    # rate 0, one pass == plain deterministic forward, std 0
    det = XceptionOracle(weights, dropout=0.0)
    m1, s1 = det.mc_predict(tiles[:1], 1, 99)
    feat = det.backbone(standardize(tiles[:1]))
    h = torch.relu(feat @ torch.from_numpy(weights['hidden_0/kernel']) + torch.from_numpy(weights['hidden_0/bias']))
    h = torch.relu(h @ torch.from_numpy(weights['hidden_1/kernel']) + torch.from_numpy(weights['hidden_1/bias']))
    p = torch.softmax(h @ torch.from_numpy(weights['logits/kernel']) + torch.from_numpy(weights['logits/bias']), 1)
    np.testing.assert_allclose(m1, p.numpy(), atol=1e-6)
    assert np.all(s1 == 0)


def test_population_std(weights):
    orc = XceptionOracle(weights)
    feat = np.abs(np.random.default_rng(0).normal(0.8, 0.5, (3, 2048))).astype(np.float32)
    m, s = orc.mc_from_features(feat, 7, 42)
    passes = np.stack([orc.head_pass(torch.from_numpy(feat), np.arange(3), p, 42).numpy().astype(np.float64)
                       for p in range(7)])
    np.testing.assert_allclose(s, passes.std(axis=0, ddof=0), atol=1e-6)
    np.testing.assert_allclose(m, passes.mean(axis=0), atol=1e-6)


def test_16bit_emulation_close(weights, tiles):
    x = standardize(tiles[:1])
    f32 = XceptionOracle(weights).backbone(x)
    bf = XceptionOracle(weights, emulate_bf16=True).backbone(x)
    assert torch.equal(bf, XceptionOracle(weights, emulate='bf16').backbone(x))       # the two spellings
    rel = ((bf - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()).item()
    assert 0 < rel < 2e-2
    h = XceptionOracle(weights, emulate='f16').backbone(x)
    rel16 = ((h - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()).item()
    assert 0 < rel16 < rel / 4                      # three more significand bits: ~8x finer
    with pytest.raises(ValueError):
        XceptionOracle(weights, emulate='fp8')


def test_f16_emulation_saturates():
    from oracle.xception_ref import _q
    t = torch.tensor([1e6, -1e6, 65504.0, 70000.0, 1.0, 6e-8])
    q = _q(t, 'f16')
    assert q.tolist()[:4] == [65504.0, -65504.0, 65504.0, 65504.0] and torch.isfinite(q).all()
    assert torch.equal(_q(t, None), t) and torch.equal(_q(t, False), t)
