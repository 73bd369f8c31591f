"""The tile reader's rows mode (bqio_decode_rows): PNG scanline filters left in for the GPU.  CPU part: the rows are what
a reference un-filter turns into the decoded tile, for every filter type; tiles that are not 8-bit RGB PNGs arrive decoded."""
import io

import numpy as np
import pytest

from biscuit_amd import tfrecord as tfr
from biscuit_amd import tfrecord_native as tn
from _png_forge import encode_png, filter_rows

pytestmark = pytest.mark.skipif(not tn.available(), reason='libbiscuit_io.so not built')
Image = pytest.importorskip('PIL.Image')


def unfilter_reference(rows):
    """[H, 1+3W] -> [H, W, 3] by the PNG specification's recurrences (plain loops: small images only)."""
    h, rs = rows.shape
    out = np.zeros((h, rs - 1), np.int64)
    for y in range(h):
        ft, cur = int(rows[y, 0]), rows[y, 1:].astype(np.int64)
        up = out[y - 1] if y else np.zeros(rs - 1, np.int64)
        for i in range(rs - 1):
            a = out[y, i - 3] if i >= 3 else 0
            b = up[i]
            c = up[i - 3] if i >= 3 else 0
            if ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            else:
                pred = (0, a, b, (a + b) >> 1)[ft]
            out[y, i] = (cur[i] + pred) & 255
    return out.astype(np.uint8).reshape(h, (rs - 1) // 3, 3)


def _write(path, payloads):
    with open(path, 'wb') as f:
        for i, img in enumerate(payloads):
            ex = tfr.encode_example('s', img, i, 0)
            import struct
            head = struct.pack('<Q', len(ex))
            f.write(head + struct.pack('<I', tfr.masked_crc(head)) + ex + struct.pack('<I', tfr.masked_crc(ex)))


def test_rows_mode_keeps_the_filters_and_a_reference_unfilter_gives_the_tile(tmp_path):
    rng = np.random.default_rng(0)
    px = 33
    img = rng.integers(0, 256, (px, px, 3), dtype=np.uint8)
    img[:, : px // 2] = (img[:, : px // 2] // 64) * 64
    pngs, kinds = [], []
    for ft in range(5):
        pngs.append(encode_png(img, np.full(px, ft)))
    pngs.append(encode_png(img, rng.integers(0, 5, px)))                 # every row its own type
    path = str(tmp_path / 'f.tfrecords')
    _write(path, pngs)
    with tn.NativeReader(path) as r:
        rows, _ = r.decode(tile_px=px, rows=True)
        full, _ = r.decode(tile_px=px)
    assert rows.shape == (6, px, 1 + 3 * px)
    for k in range(6):
        assert np.array_equal(full[k], img)                             # the host decoder on the forged files
        assert np.array_equal(rows[k], filter_rows(img, rows[k, :, 0])) # the filters are still in, byte for byte
        assert np.array_equal(unfilter_reference(rows[k]), img)
    assert [set(rows[k, :, 0]) for k in range(5)] == [{0}, {1}, {2}, {3}, {4}]


def test_rows_mode_decodes_what_is_not_rgb_png(tmp_path):
    rng = np.random.default_rng(1)
    px = 40
    img = rng.integers(0, 256, (px, px, 3), dtype=np.uint8)

    def enc(im, fmt, **kw):
        b = io.BytesIO(); im.save(b, format=fmt, **kw); return b.getvalue()
    pil = Image.fromarray(img)
    payloads = [enc(pil.convert('L'), 'PNG'), enc(pil.convert('P'), 'PNG'), enc(pil.convert('RGBA'), 'PNG'),
                enc(pil, 'JPEG', quality=90), enc(pil, 'PNG')]
    path = str(tmp_path / 'm.tfrecords')
    _write(path, payloads)
    with tn.NativeReader(path) as r:
        rows, _ = r.decode(tile_px=px, rows=True)
        full, _ = r.decode(tile_px=px)
    for k in range(4):                                                  # decoded on the host: rows of filter type 0
        assert not rows[k, :, 0].any()
        assert np.array_equal(rows[k, :, 1:].reshape(px, px, 3), full[k])
    assert np.array_equal(unfilter_reference(rows[4]), img)             # the RGB PNG keeps its filters
    # the pure-Python reader offers the same format (everything decoded, filter type 0)
    _, prow, _ = tfr.read_slide(path, px, native=False, rows=True)
    assert prow.shape == rows.shape and not prow[:, :, 0].any()
    assert np.array_equal(prow[:, :, 1:].reshape(5, px, px, 3), full)


def test_rows_mode_rejects_unknown_filter_types(tmp_path):
    px = 16
    img = np.zeros((px, px, 3), np.uint8)
    rows = filter_rows(img, np.zeros(px, np.int64))
    rows[5, 0] = 7                                                       # no such filter
    import struct, zlib

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = (b'\\x89PNG\\r\\n\\x1a\\n'.decode('unicode_escape').encode('latin1') + chunk(b'IHDR', struct.pack('>IIBBBBB', px, px, 8, 2, 0, 0, 0)) +
           chunk(b'IDAT', zlib.compress(rows.tobytes())) + chunk(b'IEND', b''))
    path = str(tmp_path / 'bad.tfrecords')
    _write(path, [png])
    with tn.NativeReader(path) as r:
        with pytest.raises(IOError):
            r.decode(tile_px=px, rows=True)
        with pytest.raises(IOError):
            r.decode(tile_px=px)
