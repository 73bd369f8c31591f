"""kernels_png.hip (`bq_png_unfilter`): the PNG scanline filters reversed on the GPU, bit for bit what a host PNG decoder
(Pillow, and the reader's own) makes of the same files -- every filter type, rows of mixed types, bands that end inside the
image (64-row bands: 299 = 4 x 64 + 43), other tile sizes -- and `evaluate` fed this way equals `evaluate` fed decoded tiles."""
import io
import struct

import numpy as np
import pytest
import torch

from biscuit_amd import tfrecord as tfr
from biscuit_amd import tfrecord_native as tn
from biscuit_amd.synthetic import make_slides
from biscuit_amd.weights import synthetic_weights
from _png_forge import encode_png

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not tn.available(), reason='libbiscuit_io.so not built')]
Image = pytest.importorskip('PIL.Image')


def _write(path, payloads):
    with open(path, 'wb') as f:
        for i, img in enumerate(payloads):
            ex = tfr.encode_example('s', img, i, 0)
            head = struct.pack('<Q', len(ex))
            f.write(head + struct.pack('<I', tfr.masked_crc(head)) + ex + struct.pack('<I', tfr.masked_crc(ex)))


def _photo(px, seed):
    r = np.random.default_rng(seed)
    y, x = np.mgrid[0:px, 0:px]
    base = np.stack([128 + 100 * np.sin(x / 17.0 + c) + 20 * np.cos(y / 9.0 * c + 1) for c in range(3)], -1)
    return np.clip(base + r.normal(0, 9, base.shape), 0, 255).astype(np.uint8)


@pytest.fixture(scope='module')
def engine():
    from biscuit_amd.engine import Engine
    e = Engine(synthetic_weights(1), dtype='f16', max_batch=32, max_mc=5)
    yield e
    e.close()


@pytest.mark.parametrize('px', [299, 64, 65, 17])
def test_every_filter_type_and_mix(engine, tmp_path, px):
    rng = np.random.default_rng(px)
    imgs = [_photo(px, 1), rng.integers(0, 256, (px, px, 3), dtype=np.uint8)]
    pngs, want = [], []
    for img in imgs:
        for types in [np.full(px, ft) for ft in range(5)] + [rng.integers(0, 5, px), rng.integers(3, 5, px)]:
            pngs.append(encode_png(img, types))
            want.append(img)
    path = str(tmp_path / 'f.tfrecords')
    _write(path, pngs)
    with tn.NativeReader(path) as r:
        rows, _ = r.decode(tile_px=px, rows=True)
    got = engine.png_unfilter(torch.from_numpy(rows).cuda()).cpu().numpy()
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g, w), (px, k)
        assert np.array_equal(w, np.asarray(Image.open(io.BytesIO(pngs[k])).convert('RGB')))    # and Pillow reads the forged file so


def test_pillow_written_tiles_and_other_kinds(engine, tmp_path):
    """What an encoder's own filter choice looks like, plus tiles the reader un-filters itself (grey, palette, JPEG)."""
    tiles = np.stack([_photo(299, s) for s in range(4)])

    def enc(im, fmt, **kw):
        b = io.BytesIO(); im.save(b, format=fmt, **kw); return b.getvalue()
    payloads = [enc(Image.fromarray(t), 'PNG') for t in tiles] + \
               [enc(Image.fromarray(tiles[0]).convert('L'), 'PNG'), enc(Image.fromarray(tiles[1]), 'JPEG', quality=90)]
    path = str(tmp_path / 'p.tfrecords')
    _write(path, payloads)
    with tn.NativeReader(path) as r:
        rows, _ = r.decode(rows=True)
        full, _ = r.decode()
    got = engine.png_unfilter(torch.from_numpy(rows).cuda()).cpu().numpy()
    assert np.array_equal(got, full)
    assert np.array_equal(got[:4], tiles)


def test_evaluate_from_filtered_rows_equals_decoded_tiles(engine, tmp_path):
    from biscuit_amd.inference import evaluate, slides_from_tfrecords
    tiles, sidx, y = make_slides(3, 7, seed=11)
    paths = []
    for i in range(3):
        p = str(tmp_path / f'u{i}.tfrecords')
        tfr.write_slide(p, f'u{i}', tiles[sidx == i])
        paths.append(p)
    labels = {f'u{i}': int(y[i]) for i in range(3)}
    a = evaluate(engine, slides_from_tfrecords(paths, labels, gpu_unfilter=True), outcome='cohort', mc_n=5, seed=3, batch=16)
    b = evaluate(engine, slides_from_tfrecords(paths, labels, gpu_unfilter=False), outcome='cohort', mc_n=5, seed=3, batch=16)
    for col in ('cohort-y_pred1', 'cohort-uncertainty1'):
        assert np.array_equal(a.tile_df[col].to_numpy(), b.tile_df[col].to_numpy()), col
    assert np.array_equal(a.slide_pred, b.slide_pred) and list(a.slide_count) == [7, 7, 7]
