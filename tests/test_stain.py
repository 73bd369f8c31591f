"""Reinhard-fast stain normaliser: oracle self-checks (CPU) and HIP kernel vs oracle (GPU).

Known answers pin the colour conversion: the CIE-LAB coordinates of the sRGB primaries under D65 are
published values (e.g. Lindbloom / scikit-image `rgb2lab`): red (53.2408, 80.0925, 67.2032),
green (87.7347, -86.1827, 83.1793), blue (32.2970, 79.1875, -107.8602), white (100, 0, 0).
The normaliser around them lives in Slideflow (not in the reference, not installable): parity unpinned.
"""
import numpy as np
import pytest

from oracle import stain

KNOWN_LAB = {
    (255, 0, 0): (53.2408, 80.0925, 67.2032),
    (0, 255, 0): (87.7347, -86.1827, 83.1793),
    (0, 0, 255): (32.2970, 79.1875, -107.8602),
    (255, 255, 255): (100.0, 0.0, 0.0),
    (0, 0, 0): (0.0, 0.0, 0.0),
}


def test_lab_known_answers():
    rgb = np.array(list(KNOWN_LAB), dtype=np.uint8)[None]          # [1, 5, 3]
    L, a, b = stain.rgb_to_lab(rgb)
    got = np.stack([L[0], a[0], b[0]], 1)
    want = np.array(list(KNOWN_LAB.values()))
    # scikit-image's rounded matrix / white point: agreement to a few 1e-2 LAB units
    assert np.abs(got - want).max() < 0.03, got


def test_lab_roundtrip_is_identity():
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (2, 37, 41, 3), dtype=np.uint8)
    L, a, b = stain.rgb_to_lab(rgb)
    back = stain.lab_to_rgb_u8(L, a, b)
    # int() truncation: a value that lands just under an integer loses one count
    d = back.astype(int) - rgb.astype(int)
    assert d.min() >= -1 and d.max() <= 0
    assert (d != 0).mean() < 0.9          # truncation (tf.cast semantics) loses a count whenever the float lands below


def _tiles(n=3, seed=0, px=299):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (n, px, px, 3)).astype(np.float32)
    tint = np.array([[0.9, 0.6, 0.8], [0.7, 0.5, 0.9], [1.0, 0.8, 0.7]], np.float32)[np.arange(n) % 3]
    smooth = np.linspace(0.6, 1.0, px, dtype=np.float32)[None, :, None, None]
    return np.clip(base * tint[:, None, None, :] * smooth, 0, 255).astype(np.uint8)


def test_fit_then_transform_matches_target_statistics():
    tiles = _tiles(3, 1, px=64)
    tm, ts = stain.fit(tiles[0])
    out = stain.reinhard_fast(tiles[1:], tm, ts)
    assert out.dtype == np.uint8 and out.shape == tiles[1:].shape
    L, a, b = stain.rgb_to_lab(out)
    mu, sd = stain.lab_stats(L, a, b)
    # clipping to the sRGB gamut and uint8 quantisation keep it from being exact
    assert np.abs(mu - tm).max() < 2.0 and np.abs(sd - ts).max() < 2.0
    # a tile normalised to its own statistics stays (nearly) unchanged
    same = stain.reinhard_fast(tiles[:1], tm, ts)
    assert np.abs(same.astype(int) - tiles[:1].astype(int)).max() <= 1


def test_constants_are_float32_inverse():
    k = stain.constants()
    assert k['lut'].dtype == np.float32 and k['lut'][0] == 0 and abs(k['lut'][255] - 1) < 1e-7
    assert np.allclose(k['m'].astype(np.float64) @ k['minv'].astype(np.float64), np.eye(3), atol=1e-6)


def _golden():
    import hashlib
    import os
    from oracle.make_stain_golden import smooth_tiles
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stain_reinhard.npz')))
    g['tiles'] = smooth_tiles(3, seed=int(g['seed']))[:2]
    assert hashlib.sha256(g['tiles'].tobytes()).digest() == g['input_sha256'].tobytes()   # same inputs as the fixture's
    return g


def test_oracle_matches_committed_fixture():
    """tests/golden/stain_reinhard.npz (oracle/make_stain_golden.py) pins the oracle's arithmetic."""
    import hashlib
    g = _golden()
    out = stain.reinhard_fast(g['tiles'], g['target_means'], g['target_stds'])
    np.testing.assert_array_equal(out[:, 100:132, 100:132], g['output_crop'])
    assert hashlib.sha256(out.tobytes()).digest() == g['output_sha256'].tobytes()
    L, a, b = stain.rgb_to_lab(g['tiles'])
    mu, sd = stain.lab_stats(L, a, b)
    np.testing.assert_array_equal(mu, g['lab_means'])
    np.testing.assert_array_equal(sd, g['lab_stds'])


@pytest.mark.gpu
def test_reinhard_kernel_matches_committed_fixture():
    import torch
    from biscuit_amd.engine import Engine
    from biscuit_amd.weights import synthetic_weights
    g = _golden()
    eng = Engine(synthetic_weights(seed=1), dtype='bf16', max_batch=8, max_mc=4)
    dt = torch.from_numpy(g['tiles']).cuda()
    got = eng.reinhard_fast(dt, g['target_means'], g['target_stds']).cpu().numpy()
    d = np.abs(got[:, 100:132, 100:132].astype(int) - g['output_crop'].astype(int))
    assert d.max() <= 1 and (d != 0).sum() <= 1
    st = eng.lab_stats(dt).cpu().numpy()
    np.testing.assert_allclose(st[:, :3], g['lab_means'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(st[:, 3:], g['lab_stds'], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_reinhard_kernel_matches_oracle():
    import torch
    from biscuit_amd.engine import Engine
    from biscuit_amd.weights import synthetic_weights
    eng = Engine(synthetic_weights(seed=1), dtype='bf16', max_batch=8, max_mc=4)
    tiles = _tiles(5, 7)
    tm, ts = stain.fit(tiles[0])
    want = stain.reinhard_fast(tiles, tm, ts)
    dt = torch.from_numpy(tiles).cuda()
    got = eng.reinhard_fast(dt, tm, ts).cpu().numpy()
    d = np.abs(got.astype(int) - want.astype(int))
    # same precision contract on both sides: identical up to rare double-rounding flips of one count
    assert d.max() <= 1 and (d != 0).mean() < 1e-5, (d.max(), (d != 0).mean())
    # statistics (fit) and the in-place form
    st = eng.lab_stats(dt).cpu().numpy()
    L, a, b = stain.rgb_to_lab(tiles)
    mu, sd = stain.lab_stats(L, a, b)
    np.testing.assert_allclose(st[:, :3], mu, rtol=0, atol=1e-5)
    np.testing.assert_allclose(st[:, 3:], sd, rtol=0, atol=1e-5)
    inplace = dt.clone()
    eng.reinhard_fast(inplace, tm, ts, out=inplace)
    assert torch.equal(inplace.cpu(), torch.from_numpy(got))


@pytest.mark.gpu
def test_normaliser_object_and_pipeline():
    import torch
    from biscuit_amd.engine import Engine, UncertaintyInterface
    from biscuit_amd.inference import Slide, evaluate
    from biscuit_amd.stain import ReinhardFast
    from biscuit_amd.weights import synthetic_weights
    from oracle.xception_ref import XceptionOracle
    weights = synthetic_weights(seed=1)
    eng = Engine(weights, dtype='f32', max_batch=8, max_mc=4)
    tiles = _tiles(4, 11)
    norm = ReinhardFast(eng).fit(tiles[0])
    tm, ts = stain.fit(tiles[0])
    fit = norm.get_fit()
    np.testing.assert_allclose(fit['target_means'], tm, atol=1e-5)
    np.testing.assert_allclose(fit['target_stds'], ts, atol=1e-5)
    one = norm.rgb_to_rgb(tiles[1])                       # single image, like results.py:252
    assert tuple(one.shape) == (299, 299, 3) and one.dtype == torch.uint8
    with pytest.raises(ValueError):
        ReinhardFast.from_params(eng, {'norm': 'reinhard_fast'})
    assert UncertaintyInterface(eng, uq_n=2, norm_fit=fit).wsi_normalizer is not None
    # end to end: evaluate(norm_fit=...) == oracle normalise -> oracle MC inference
    slides = [Slide('s0', torch.from_numpy(tiles[:2]), 2, y_true=0), Slide('s1', torch.from_numpy(tiles[2:]), 2, y_true=1)]
    res = evaluate(eng, slides, mc_n=3, seed=5, batch=4, norm_fit=fit)
    ref_tiles = stain.reinhard_fast(tiles, tm, ts)
    rm, rs = XceptionOracle(weights).mc_predict(ref_tiles, 3, 5, mode='head')
    df = res.tile_df
    assert np.abs(df['cohort-y_pred1'].to_numpy() - rm[:, 1]).max() < 1e-4
    assert np.abs(df['cohort-uncertainty1'].to_numpy() - rs[:, 1]).max() < 1e-4
