"""libbiscuit_io.so (include/biscuit_io.h): native TFRecord framing / Example parse / PNG decode against
the pure-Python reader and Pillow on self-written records (there are no real TFRecords here)."""
import io
import os
import re
import struct

import numpy as np
import pytest

from biscuit_amd import tfrecord as tfr
from biscuit_amd import tfrecord_native as tn

pytestmark = pytest.mark.skipif(not tn.available(), reason='libbiscuit_io.so not built')


def _tiles(n, seed=0, px=299):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (n, px, px, 3), dtype=np.uint8)
    t[:, :, : px // 2] = (t[:, :, : px // 2] // 32) * 32          # smooth areas: every PNG filter type gets used
    t[:, : px // 3] = np.linspace(0, 255, px, dtype=np.uint8)[None, None, :, None]
    return t


def test_header_symbols_exported():
    text = open(os.path.join(os.path.dirname(tn.LIB_PATH), '..', 'include', 'biscuit_io.h')).read()
    declared = set(re.findall(r'\b(bqio_[a-z0-9_]+)\s*\(', text))
    assert declared == set(tn.ABI), declared ^ set(tn.ABI)
    tn.load()        # every declared symbol resolves


def test_crc_matches_python():
    for data in (b'', b'a', b'123456789', bytes(range(256)) * 7 + b'xyz'):
        assert tn.lib().bqio_masked_crc32c(data, len(data)) == tfr.masked_crc(data)
    assert tfr.crc32c(b'123456789') == 0xE3069283     # CRC-32C check value (RFC 3720)


def test_native_equals_python(tmp_path):
    tiles = _tiles(7, 1)
    locs = np.arange(14, dtype=np.int64).reshape(7, 2) * 302 - 5
    path = str(tmp_path / 's1.tfrecords')
    tfr.write_slide(path, 'slide-1', tiles, locs)
    with tn.NativeReader(path, verify='full') as r:
        assert len(r) == 7 and r.slide == 'slide-1'
        assert r.image_format(0) == tn.IMG_PNG
        assert r.image_bytes(3)[:8] == b'\x89PNG\r\n\x1a\n'
        got, gl = r.decode(threads=3)
        np.testing.assert_array_equal(got, tiles)
        np.testing.assert_array_equal(gl, locs)
        part, pl = r.decode(2, 3, threads=1)
        np.testing.assert_array_equal(part, tiles[2:5])
        np.testing.assert_array_equal(pl, locs[2:5])
        buf = np.zeros((7, 299, 299, 3), np.uint8)              # caller-owned (pinned) buffer
        out, _ = r.decode(out=buf)
        assert out is buf and np.array_equal(buf, tiles)
    name, t2, l2 = tfr.read_slide(path, native=True)
    name3, t3, l3 = tfr.read_slide(path, native=False)
    assert name == name3 == 'slide-1'
    np.testing.assert_array_equal(t2, t3)
    np.testing.assert_array_equal(l2, l3)


def _png(arr, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format='PNG', **kw)
    return b.getvalue()


def _write(path, payloads, slide='s'):
    with open(path, 'wb') as f:
        for raw in payloads:
            rec = tfr.encode_example(slide, raw, 1, 2)
            head = struct.pack('<Q', len(rec))
            f.write(head + struct.pack('<I', tfr.masked_crc(head)) + rec + struct.pack('<I', tfr.masked_crc(rec)))


def test_png_colour_types_and_filters(tmp_path):
    from PIL import Image
    rgb = _tiles(1, 2)[0]
    grey = rgb[..., 0]
    rgba = np.dstack([rgb, np.full(rgb.shape[:2], 200, np.uint8)])
    pal = Image.fromarray(rgb).quantize(64)
    b = io.BytesIO()
    pal.save(b, format='PNG')
    payloads = [_png(rgb, compress_level=1), _png(rgb, optimize=True), _png(grey), _png(rgba), b.getvalue()]
    path = str(tmp_path / 'types.tfrecords')
    _write(path, payloads)
    want = np.stack([rgb, rgb, np.repeat(grey[..., None], 3, 2), rgb, np.asarray(pal.convert('RGB'))])
    with tn.NativeReader(path) as r:
        got, _ = r.decode()
    np.testing.assert_array_equal(got, want)


def test_errors(tmp_path):
    tiles = _tiles(2, 3)
    path = str(tmp_path / 'ok.tfrecords')
    tfr.write_slide(path, 'x', tiles)
    raw = bytearray(open(path, 'rb').read())
    bad = str(tmp_path / 'bad.tfrecords')
    raw[40] ^= 0xFF                                   # payload byte of record 0
    open(bad, 'wb').write(raw)
    tn.NativeReader(bad, verify='length').close()     # header CRCs still fine
    with pytest.raises(IOError, match='corrupt record data'):
        tn.NativeReader(bad, verify='full')
    raw = bytearray(open(path, 'rb').read())
    raw[3] ^= 0x01                                    # length field
    open(bad, 'wb').write(raw)
    with pytest.raises(IOError):
        tn.NativeReader(bad, verify='length')
    open(bad, 'wb').write(open(path, 'rb').read()[:-9])
    with pytest.raises(IOError, match='truncated'):
        tn.NativeReader(bad)
    with pytest.raises(IOError):
        tn.NativeReader(str(tmp_path / 'missing.tfrecords'))
    # wrong tile size, JPEG payload, empty file
    small = str(tmp_path / 'small.tfrecords')
    _write(small, [_png(tiles[0][:100, :100])])
    with tn.NativeReader(small) as r, pytest.raises(ValueError, match='tile size'):
        r.decode()
    from PIL import Image
    jb = io.BytesIO()
    Image.fromarray(tiles[0]).save(jb, format='JPEG', quality=95)
    mixed = str(tmp_path / 'mixed.tfrecords')
    _write(mixed, [_png(tiles[0]), jb.getvalue()])
    with tn.NativeReader(mixed) as r:
        assert r.image_format(1) == tn.IMG_JPEG
        with pytest.raises(tn.UnsupportedImage) as ei:
            r.decode()
        assert ei.value.index == 1
    name, t, _ = tfr.read_slide(mixed)               # falls back to Pillow for the JPEG record
    assert t.shape == (2, 299, 299, 3) and np.array_equal(t[0], tiles[0])
    empty = str(tmp_path / 'empty.tfrecords')
    open(empty, 'wb').close()
    with tn.NativeReader(empty) as r:
        assert len(r) == 0 and r.slide is None
    assert tfr.read_slide(empty)[1].shape == (0, 299, 299, 3)


def test_mutated_files_never_crash(tmp_path):
    """Byte-mutated / truncated records must come back as errors, not as a crash of the process
    (the reader parses untrusted files in C++): decode them in a child process and check it survives."""
    import random
    import subprocess
    import sys
    tiles = _tiles(2, 5)
    path = str(tmp_path / 'a.tfrecords')
    tfr.write_slide(path, 's', tiles)
    raw = open(path, 'rb').read()
    rnd = random.Random(7)
    paths = []
    for k in range(120):
        b = bytearray(raw)
        for _ in range(rnd.choice([1, 2, 5, 20])):
            i = rnd.randrange(400) if rnd.random() < 0.4 else rnd.randrange(len(b))
            b[i] = rnd.randrange(256)
        if rnd.random() < 0.2:
            b = b[:rnd.randrange(len(b))]
        q = str(tmp_path / f'm{k}.tfrecords')
        open(q, 'wb').write(b)
        paths.append(q)
    code = (
        "import sys\n"
        "from biscuit_amd import tfrecord_native as tn\n"
        "for p in sys.argv[1:]:\n"
        "    try:\n"
        "        with tn.NativeReader(p, verify=None) as r:\n"
        "            for i in range(len(r)):\n"
        "                try: r.image_format(i); r.image_bytes(i)\n"
        "                except Exception: pass\n"
        "            try: r.slide; r.decode(threads=2)\n"
        "            except Exception: pass\n"
        "    except Exception: pass\n"
        "print('survived')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(tn.LIB_PATH)))
    out = subprocess.run([sys.executable, '-c', code] + paths, capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0 and 'survived' in out.stdout, out.stderr[-500:]
