"""Slideflow-format TFRecord reader on self-written records: there are no real
TFRecords and no TensorFlow here, so the pin is the published wire format + CRC-32C check value."""
import struct

import numpy as np
import pytest

from biscuit_amd import tfrecord as T
from biscuit_amd.synthetic import make_tiles


def test_crc32c_check_value():
    assert T.crc32c(b'123456789') == 0xE3069283          # CRC-32C (Castagnoli) check value
    assert T.crc32c(b'') == 0
    assert T.masked_crc(b'\x00' * 8) == ((((T.crc32c(b'\x00' * 8) >> 15) | (T.crc32c(b'\x00' * 8) << 17))
                                          & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_png_roundtrip_exact(tmp_path):
    tiles = make_tiles(3, seed=9)
    locs = [(10, 20), (300, -5), (2 ** 40, 7)]
    path = str(tmp_path / 'slideA.tfrecords')
    T.write_slide(path, 'slideA', tiles, locs, fmt='PNG')
    name, got, loc = T.read_slide(path, verify='full')
    assert name == 'slideA'
    assert got.dtype == np.uint8 and np.array_equal(got, tiles)          # PNG is lossless
    assert loc.tolist() == [list(x) for x in locs]
    feats = T.parse_example(next(T.read_records(path)))
    assert set(feats) == {'image_raw', 'loc_x', 'loc_y', 'slide'} and feats['image_raw'][:4] == b'\x89PNG'


def test_jpeg_and_empty(tmp_path):
    tiles = make_tiles(2, seed=3)
    path = str(tmp_path / 's.tfrecords')
    T.write_slide(path, 's', tiles, fmt='JPEG')
    _, got, _ = T.read_slide(path)
    # lossy: noisy synthetic tiles + chroma subsampling; just check it is the same picture
    assert got.shape == tiles.shape and np.abs(got.astype(int) - tiles.astype(int)).mean() < 20
    assert np.corrcoef(got.ravel().astype(float), tiles.ravel().astype(float))[0, 1] > 0.9
    empty = str(tmp_path / 'e.tfrecords')
    open(empty, 'wb').close()
    name, t, loc = T.read_slide(empty)
    assert name is None and t.shape == (0, 299, 299, 3) and loc.shape == (0, 2)


def test_corruption_is_detected(tmp_path):
    tiles = make_tiles(1, seed=1)
    path = str(tmp_path / 'c.tfrecords')
    T.write_slide(path, 'c', tiles)
    raw = bytearray(open(path, 'rb').read())
    bad = bytearray(raw); bad[3] ^= 1                      # length field
    open(path, 'wb').write(bad)
    with pytest.raises(IOError):
        list(T.read_records(path))
    bad = bytearray(raw); bad[40] ^= 1                     # payload byte
    open(path, 'wb').write(bad)
    list(T.read_records(path, verify='length'))            # header-only check passes
    with pytest.raises(IOError):
        list(T.read_records(path, verify='full'))
    open(path, 'wb').write(raw[:-7])                       # truncated
    with pytest.raises(IOError):
        list(T.read_records(path))
    wrong = str(tmp_path / 'w.tfrecords')
    T.write_slide(wrong, 'w', np.zeros((1, 64, 64, 3), np.uint8))
    with pytest.raises(ValueError):
        T.read_slide(wrong)


def test_evaluate_from_tfrecords(tmp_path):
    """The inference driver over TFRecord slides (device engine replaced by the CPU stand-in)."""
    from biscuit_amd.inference import evaluate, slides_from_tfrecords
    from tests.test_distributed import StandInEngine
    paths = []
    for i, n in enumerate((3, 0, 2)):
        p = str(tmp_path / f'slide{i}.tfrecords')
        T.write_slide(p, f'slide{i}', make_tiles(n, seed=20 + i)) if n else open(p, 'wb').close()
        paths.append(p)
    assert [T.count_records(p) for p in paths] == [3, 0, 2]
    slides = slides_from_tfrecords(paths, {'slide0': 1, 'slide2': 0})
    res = evaluate(StandInEngine(), slides, outcome='cohort', mc_n=30, seed=1, batch=4)
    assert list(res.slide_count) == [3, 0, 2] and len(res.tile_df) == 5
    assert list(res.tile_df['slide']) == ['slide0'] * 3 + ['slide2'] * 2
    assert list(res.tile_df['cohort-y_true0']) == [1, 1, 1, 0, 0]
