"""Slideflow tile TFRecords -> uint8 tile arrays for the staging kernel.

The reference stores 299 px / 302 um tiles as PNG inside ``*.tfrecords``, one file per slide
(``configure.py:118-124`` ``img_format='png'``).  Neither TensorFlow nor Slideflow is needed to read
them: the container is length-prefixed records
    uint64 length | uint32 masked_crc32c(length) | bytes[length] | uint32 masked_crc32c(data)
each holding a ``tf.train.Example`` protobuf with features ``slide`` (bytes), ``image_raw``
(bytes: PNG or JPEG), ``loc_x`` / ``loc_y`` (int64).  Only that subset of protobuf is parsed.
PNG and baseline-JPEG tiles are decoded by the native reader (libbiscuit_io.so: C++ framing, protobuf subset, its own
inflate + unfilter and its own JPEG decoder, a pool of host threads; include/biscuit_io.h); Pillow decodes what that
reader refuses (progressive or damaged JPEG) and is the fallback when the library is not built.  A writer for the same wire format is included so the
reader can be tested without TensorFlow (there are no real TFRecords in this environment).
"""
import io
import struct

import numpy as np

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78                      # CRC-32C (Castagnoli), reflected
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_NATIVE_CRC = None


def masked_crc(data: bytes) -> int:
    """TFRecord's masked CRC-32C.  Payloads go through libbiscuit_io when it is there (slicing-by-8: a 227 KB PNG record costs
    0.1 ms instead of the 20 ms of the byte loop above -- writing the benchmark's 16 384 records took three minutes of it)."""
    global _NATIVE_CRC
    if len(data) >= 64:
        if _NATIVE_CRC is None:
            try:
                from . import tfrecord_native
                _NATIVE_CRC = tfrecord_native.lib().bqio_masked_crc32c if tfrecord_native.available() else False
            except Exception:                                    # noqa: BLE001 -- the pure-Python form below always works
                _NATIVE_CRC = False
        if _NATIVE_CRC:
            return int(_NATIVE_CRC(bytes(data), len(data)))
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- minimal protobuf -------------------------------------------------------------
def _varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message; value is int or a memoryview slice."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fn, wt, v


def parse_example(payload):
    """tf.train.Example bytes -> {name: bytes | [int] | [float]}."""
    out = {}
    for fn, _, features in _fields(memoryview(payload)):
        if fn != 1:
            continue
        for fn2, _, entry in _fields(features):
            if fn2 != 1:
                continue
            key, feat = None, None
            for fn3, _, v in _fields(entry):
                if fn3 == 1:
                    key = bytes(v).decode()
                elif fn3 == 2:
                    feat = v
            if key is None or feat is None:
                continue
            for kind, _, lst in _fields(feat):
                if kind == 1:        # BytesList
                    vals = [bytes(v) for f, _, v in _fields(lst) if f == 1]
                    out[key] = vals[0] if len(vals) == 1 else vals
                elif kind == 3:      # Int64List (packed or not)
                    vals = []
                    for f, wt, v in _fields(lst):
                        if f != 1:
                            continue
                        if wt == 0:
                            vals.append(v)
                        else:
                            j = 0
                            while j < len(v):
                                x, j = _varint(v, j)
                                vals.append(x)
                    out[key] = [x - (1 << 64) if x >> 63 else x for x in vals]
                elif kind == 2:      # FloatList (packed)
                    vals = []
                    for f, wt, v in _fields(lst):
                        if f == 1:
                            vals += list(struct.unpack(f'<{len(v) // 4}f', bytes(v)))
                    out[key] = vals
    return out


def read_records(path, verify='length'):
    """Yield raw record payloads.  verify: None, 'length' (header CRC only) or 'full'."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError(f'{path}: truncated record header')
            (length,), (lcrc,) = struct.unpack('<Q', head[:8]), struct.unpack('<I', head[8:])
            if verify and masked_crc(head[:8]) != lcrc:
                raise IOError(f'{path}: corrupt record length')
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError(f'{path}: truncated record')
            if verify == 'full' and masked_crc(data) != struct.unpack('<I', tail)[0]:
                raise IOError(f'{path}: corrupt record data')
            yield data


def count_records(path):
    """Number of records (tiles) in a TFRecord file, reading only the 12-byte headers."""
    n = 0
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return n
            if len(head) < 12:
                raise IOError(f'{path}: truncated record header')
            f.seek(struct.unpack('<Q', head[:8])[0] + 4, 1)
            n += 1


def decode_image(raw, tile_px=299):
    from PIL import Image
    img = np.asarray(Image.open(io.BytesIO(raw)).convert('RGB'))
    if img.shape != (tile_px, tile_px, 3):
        raise ValueError(f'tile is {img.shape}, expected {(tile_px, tile_px, 3)}')
    return img


def read_slide(path, tile_px=299, verify='length', native=None, out=None, threads=None, rows=False):
    """One slide's TFRecord -> (slide name, tiles uint8 [T,px,px,3], loc int64 [T,2]).

    ``rows=True``: tiles as [T,px,1+3*px] -- per row a PNG filter-type byte and the (still filtered) RGB bytes, for
    ``Engine.png_unfilter`` on the GPU; whatever is not an 8-bit RGB PNG arrives decoded, as rows of filter type 0.

    ``native=None`` uses libbiscuit_io.so (C++ framing / protobuf / PNG decode on a thread pool) when it
    is built and falls back to the pure-Python reader otherwise; a slide with a record the native decoder
    refuses (progressive JPEG, a damaged stream) is decoded with Pillow either way.  ``out``: optional uint8 buffer [T,px,px,3] to
    decode into (pinned memory for an overlapped H2D copy)."""
    if native is None:
        from . import tfrecord_native
        native = tfrecord_native.available()
    if native:
        from . import tfrecord_native as tn
        with tn.NativeReader(path, verify) as r:
            n = len(r)
            if n == 0:
                return r.slide, np.zeros((0, tile_px, 1 + 3 * tile_px) if rows else (0, tile_px, tile_px, 3), np.uint8), \
                    np.zeros((0, 2), np.int64)
            try:
                tiles, locs = r.decode(0, n, tile_px, out=out, threads=threads, rows=rows)
                return r.slide, tiles, locs
            except tn.UnsupportedImage:
                name = r.slide            # outside the native decoders' subset: Pillow below
    tiles, locs, name = [], [], None
    for payload in read_records(path, verify):
        ex = parse_example(payload)
        if name is None and 'slide' in ex:
            name = ex['slide'].decode()
        tiles.append(decode_image(ex['image_raw'], tile_px))
        locs.append((ex.get('loc_x', [0])[0], ex.get('loc_y', [0])[0]))
    if not tiles:
        return name, np.zeros((0, tile_px, 1 + 3 * tile_px) if rows else (0, tile_px, tile_px, 3), np.uint8), \
            np.zeros((0, 2), np.int64)
    tiles = np.stack(tiles)
    if rows:                                    # decoded here: rows of filter type 0
        plain = np.zeros((len(tiles), tile_px, 1 + 3 * tile_px), np.uint8)
        plain[:, :, 1:] = tiles.reshape(len(tiles), tile_px, 3 * tile_px)
        tiles = plain
    if out is not None:
        out[...] = tiles
        tiles = out
    return name, tiles, np.asarray(locs, dtype=np.int64)


# ---- writer (tests / synthetic datasets) ---------------------------------------------
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_example(slide, image_raw, loc_x, loc_y):
    def feature_bytes(b):
        return _ld(1, _ld(1, b))
    def feature_int(v):
        return _ld(3, _ld(1, _enc_varint(v)))
    entries = b''
    for key, feat in (('image_raw', feature_bytes(image_raw)), ('loc_x', feature_int(loc_x)),
                      ('loc_y', feature_int(loc_y)), ('slide', feature_bytes(slide.encode()))):
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)


def encode_image(tile, fmt='PNG'):
    """One tile as the bytes Slideflow stores in ``image_raw``."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.asarray(tile, np.uint8)).save(buf, format=fmt, **({'quality': 95} if fmt == 'JPEG' else {}))
    return buf.getvalue()


def write_slide(path, slide, tiles, locs=None, fmt='PNG'):
    """One TFRecord file for one slide.  ``tiles``: arrays, or already encoded images (bytes)."""
    with open(path, 'wb') as f:
        for i, t in enumerate(tiles):
            image = bytes(t) if isinstance(t, (bytes, bytearray, memoryview)) else encode_image(t, fmt)
            lx, ly = (locs[i] if locs is not None else (i, 0))
            payload = encode_example(slide, image, int(lx), int(ly))
            head = struct.pack('<Q', len(payload))
            f.write(head + struct.pack('<I', masked_crc(head)) + payload + struct.pack('<I', masked_crc(payload)))
