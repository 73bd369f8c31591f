"""The tile reader's baseline-JPEG decoder (biscuit_amd/csrc/jpeg_baseline.h, `bqio_decode_jpeg` / `bqio_decode`) against
Pillow, i.e. against libjpeg-turbo with the defaults TensorFlow's decode_jpeg also uses (integer-accurate IDCT, fancy
upsampling): the same bytes, for every sampling, quality and table layout an encoder here can produce; streams outside
the subset, and damaged ones, are refused so that the caller's fallback decides."""
import io
import random

import numpy as np
import pytest

from biscuit_amd import tfrecord as tfr
from biscuit_amd import tfrecord_native as tn

Image = pytest.importorskip('PIL.Image')
pytestmark = pytest.mark.skipif(not tn.available(), reason='libbiscuit_io.so not built')


def _photo(px, seed=0):
    """Smooth gradients + grain: long zero runs, short codes and the DC-only fast path all get used."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:px, 0:px]
    base = np.stack([128 + 100 * np.sin(x / 17.0 + c) + 20 * np.cos(y / 9.0 * c + 1) for c in range(3)], -1)
    base[: px // 4] = 200                                                       # a flat band
    return np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)


def _noise(px, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (px, px, 3), dtype=np.uint8)


def _enc(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, format='JPEG', **kw)
    return b.getvalue()


def _pillow(raw):
    return np.asarray(Image.open(io.BytesIO(raw)).convert('RGB'))


@pytest.mark.parametrize('px', [299, 300, 64, 33, 17])
def test_same_bytes_as_libjpeg(px):
    """4:4:4, 4:2:2 and 4:2:0; qualities from coarse to lossless-ish; default and optimised Huffman tables; sizes that
    are and are not multiples of the MCU (edge blocks, odd chroma widths, the last chroma row's replication)."""
    n = 0
    for img in (_photo(px), _noise(px, 1)):
        for q in (30, 75, 95, 100):
            for ss in (0, 1, 2):
                for opt in (False, True):
                    try:
                        raw = _enc(img, quality=q, subsampling=ss, optimize=opt)
                    except OSError:          # Pillow's own buffer limit on tiny, incompressible inputs
                        continue
                    assert np.array_equal(tn.decode_jpeg(raw, px), _pillow(raw)), (px, q, ss, opt)
                    n += 1
    assert n >= 40


def test_grey_restarts_and_saturated_colours():
    img = _photo(299, 2)
    raw = _enc(img[..., 0], quality=90)                                         # one component
    assert np.array_equal(tn.decode_jpeg(raw), _pillow(raw))
    for kw in (dict(restart_marker_blocks=5), dict(restart_marker_rows=1), dict(restart_marker_blocks=1)):
        for ss in (0, 1, 2):
            raw = _enc(img, quality=85, subsampling=ss, **kw)
            assert raw.count(b'\xff\xd0') > 0
            assert np.array_equal(tn.decode_jpeg(raw), _pillow(raw)), (kw, ss)
    # primaries and their complements in hard-edged blocks: every clamp of the colour conversion is hit
    sat = np.zeros((299, 299, 3), np.uint8)
    cols = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (0, 255, 255), (255, 0, 255), (255, 255, 255), (0, 0, 0)]
    for i in range(299 // 13 + 1):
        for j in range(299 // 13 + 1):
            sat[13 * i: 13 * i + 13, 13 * j: 13 * j + 13] = cols[(3 * i + j) % 8]
    for ss in (0, 1, 2):
        raw = _enc(sat, quality=60, subsampling=ss)
        assert np.array_equal(tn.decode_jpeg(raw), _pillow(raw)), ss


def test_outside_the_subset_is_refused():
    img = _photo(299, 3)
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(_enc(img, quality=85, progressive=True))
    raw = _enc(img, quality=85)
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(raw[: len(raw) // 2])                                    # cut off
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(raw[: len(raw) // 2] + b'\xff\xd9')                      # cut off, then a tidy end marker
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(raw.replace(b'\xff\xc0', b'\xff\xc9', 1))              # frame marker of an arithmetic-coded file
    cmyk = io.BytesIO()
    Image.fromarray(img).convert('CMYK').save(cmyk, format='JPEG', quality=85)
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(cmyk.getvalue())                                         # four components
    with pytest.raises(ValueError):
        tn.decode_jpeg(raw, 298)                                                # a tile of another size
    with pytest.raises(tn.UnsupportedImage):
        tn.decode_jpeg(b'\xff\xd8\xff')


def test_scan_cut_to_nothing_before_a_valid_end_marker_is_refused():
    """Round-3 advisory: a valid header, an entropy-coded segment far shorter than its MCU count and an EOI.  The bit reader
    refills eight bytes at a time and used to be checked once per restart interval only: it walked hundreds of bytes per block
    past the unstuffed scan.  Now every block starts with a bound check; the file is UNSUPPORTED (Pillow fallback), for every
    remaining scan length and in the middle of a slide's records as well."""
    for ss, kw in ((0, {}), (2, {}), (2, {'restart_marker_blocks': 7})):
        try:
            raw = _enc(_photo(299, 5), quality=90, subsampling=ss, **kw)
        except TypeError:
            continue
        sos = raw.index(b'\xff\xda')
        hdr = sos + 2 + int.from_bytes(raw[sos + 2:sos + 4], 'big')
        for keep in (0, 1, 2, 7, 8, 9, 33, 500):
            body = raw[hdr:hdr + keep].replace(b'\xff', b'\x7f')
            with pytest.raises(tn.UnsupportedImage):
                tn.decode_jpeg(raw[:hdr] + body + b'\xff\xd9')
        with pytest.raises(tn.UnsupportedImage):
            tn.decode_jpeg(raw[:hdr] + b'\x00\x00\xff\xd9')                        # the advisory's reproducer


def test_damaged_streams_agree_or_are_refused():
    """Byte-flipped files: whatever the native decoder accepts must still equal libjpeg's output (it refuses as soon as a
    stream leaves the arithmetic range in which libjpeg's builds agree with each other), and nothing may crash."""
    raw = _enc(_photo(299, 4), quality=85)
    rnd = random.Random(7)
    accepted = 0
    for _ in range(600):
        b = bytearray(raw)
        for _ in range(rnd.randint(1, 4)):
            b[rnd.randrange(2, len(b))] = rnd.randrange(256)
        try:
            got = tn.decode_jpeg(bytes(b))
        except (tn.UnsupportedImage, ValueError):
            continue
        accepted += 1
        assert np.array_equal(got, _pillow(bytes(b)))
    assert accepted > 100


def test_jpeg_slide_through_the_reader(tmp_path):
    """A JPEG TFRecord (Slideflow's img_format='jpg') through read_slide: decoded by the native thread pool, tiles equal
    to Pillow's, locations intact, for every thread count."""
    tiles = np.stack([_photo(299, s) for s in range(5)])
    locs = np.arange(10, dtype=np.int64).reshape(5, 2) * 151
    path = str(tmp_path / 'j.tfrecords')
    tfr.write_slide(path, 'slide-j', tiles, locs, fmt='JPEG')
    want = np.stack([_pillow(tfr.parse_example(p)['image_raw']) for p in tfr.read_records(path)])
    for threads in (1, 2, 8):
        with tn.NativeReader(path) as r:
            assert r.image_format(0) == tn.IMG_JPEG
            got, loc = r.decode(threads=threads)
        assert np.array_equal(got, want) and np.array_equal(loc, locs)
    name, got, loc = tfr.read_slide(path)
    assert name == 'slide-j' and np.array_equal(got, want) and np.array_equal(loc, locs)
    _, py, _ = tfr.read_slide(path, native=False)
    assert np.array_equal(py, want)


def test_one_decoder_per_slide_decided_before_the_first_chunk(tmp_path):
    """A slide whose LAST record is a progressive JPEG: ``NativeReader.probe`` names it without decoding anything, and the chunk
    source of ``evaluate`` sends the whole slide to the fallback decoder from chunk 0 on -- not the chunks from that record on, which
    would mix two decoders' IDCTs inside one slide (and differ from the whole-slide loader)."""
    from biscuit_amd.inference import TFRecordSource
    imgs = [_photo(299, s) for s in range(6)]
    raws = [_enc(a, quality=85) for a in imgs[:5]] + [_enc(imgs[5], quality=85, progressive=True)]
    path = str(tmp_path / 'mixed.tfrecords')
    tfr.write_slide(path, 'mixed', raws, np.zeros((6, 2), np.int64))
    good = str(tmp_path / 'good.tfrecords')
    tfr.write_slide(good, 'good', raws[:5], np.zeros((5, 2), np.int64))
    with tn.NativeReader(path) as r:
        assert r.probe(299) == 5 and r.probe(299, 0, 5) is None
    with tn.NativeReader(good) as r:
        assert r.probe(299) is None
    want = np.stack([_pillow(x) for x in raws])
    src = TFRecordSource(path, 6)
    out = np.zeros((2, 299, 299, 3), np.uint8)
    src.read(0, 2, out)                                 # the first chunk already comes from the fallback
    assert src._reader is None and src._fallback is not None and np.array_equal(out, want[:2])
    src.read(4, 2, out)
    assert np.array_equal(out, want[4:6])
    src.close()
    src = TFRecordSource(good, 5)
    src.read(0, 2, out)
    assert src._reader is not None and src._fallback is None and np.array_equal(out, want[:2])
    src.close()
