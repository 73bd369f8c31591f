"""ctypes binding of libbiscuit_io.so (include/biscuit_io.h): the native TFRecord / PNG reader.

`NativeReader(path)` indexes one slide's TFRecord; `decode(first, count)` returns uint8 tiles
``[count, px, px, 3]`` decoded by a pool of host threads, optionally straight into a caller-supplied
(e.g. pinned) buffer so the H2D copy of one batch overlaps the decode of the next.  PNG and baseline JPEG
payloads are decoded natively; for anything else (progressive JPEG, a damaged stream) `decode` raises
`UnsupportedImage` and the caller (tfrecord.read_slide) decodes the slide with Pillow.
"""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libbiscuit_io.so')

VERIFY = {None: 0, False: 0, 'none': 0, 'length': 1, 'full': 2}
IMG_PNG, IMG_JPEG = 1, 2
ERR_UNSUPPORTED = -5
ERR_FORMAT = -3

_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
ABI = {
    'bqio_open': (_vp, [C.c_char_p, _i]),
    'bqio_close': (None, [_vp]),
    'bqio_last_error': (C.c_char_p, [_vp]),
    'bqio_count': (_i64, [_vp]),
    'bqio_slide_name': (_i, [_vp, C.c_char_p, _i]),
    'bqio_image_format': (_i, [_vp, _i64]),
    'bqio_image_bytes': (_i, [_vp, _i64, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    'bqio_decode': (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, C.POINTER(_i64)]),
    'bqio_decode_rows': (_i, [_vp, _i64, _i64, _i, _vp, _vp, _i, C.POINTER(_i64)]),
    'bqio_probe': (_i, [_vp, _i64, _i64, _i, C.POINTER(_i64)]),
    'bqio_extract_z': (_i, [_vp, _i64, _i64, _i, _vp, C.c_size_t, _vp, _vp, _vp, C.POINTER(C.c_size_t), _i, C.POINTER(_i64)]),
    'bqio_masked_crc32c': (C.c_uint32, [C.c_char_p, C.c_size_t]),
    'bqio_inflate': (_i, [C.c_char_p, C.c_size_t, _vp, C.c_size_t]),
    'bqio_inflate2': (_i, [C.c_char_p, C.c_size_t, _vp, C.c_size_t, C.c_char_p, C.c_size_t, _vp, C.c_size_t,
                          C.POINTER(_i), C.POINTER(_i)]),
    'bqio_inflate_fallbacks': (_i64, []),
    'bqio_decode_jpeg': (_i, [C.c_char_p, C.c_size_t, _i, _vp]),
    # the output side: the tile-prediction table (biscuit_amd/predictions.py)
    'bqio_table_open': (_vp, [C.c_char_p, C.c_char_p, _i, _i]),
    'bqio_table_last_error': (C.c_char_p, [_vp]),
    'bqio_table_rows': (_i, [_vp, C.c_char_p, _i64, _vp, _vp, _vp, _i64]),
    'bqio_table_tell': (_i64, [_vp]),
    'bqio_table_append_file': (_i, [_vp, C.c_char_p, _i64, _i64]),
    'bqio_table_close': (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    'bqio_format_f64': (_i, [C.c_double, C.c_char_p, _i]),
}


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(f'{path} not found; build it with `make -C biscuit_amd/csrc`')
    lib = C.CDLL(path)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def available():
    return os.path.exists(LIB_PATH)


def inflate(zdata, out_len):
    """The reader's own zlib-stream decompressor (csrc/inflate_fast.h): ``zlib.decompress(zdata)`` when that has exactly
    ``out_len`` bytes, ValueError for anything zlib would refuse.  For tests."""
    import numpy as np
    out = np.empty(out_len, np.uint8)
    e = lib().bqio_inflate(bytes(zdata), len(zdata), out.ctypes.data, out_len)
    if e != 0:
        raise ValueError(f'bqio_inflate: error {e}')
    return out.tobytes()


def inflate2(za, len_a, zb, len_b):
    """Two zlib streams through the reader's two-stream loop: (bytes | None, bytes | None), None where ``inflate`` would
    raise.  For tests."""
    import numpy as np
    oa, ob = np.empty(len_a, np.uint8), np.empty(len_b, np.uint8)
    ka, kb = C.c_int(0), C.c_int(0)
    e = lib().bqio_inflate2(bytes(za), len(za), oa.ctypes.data, len_a, bytes(zb), len(zb), ob.ctypes.data, len_b,
                            C.byref(ka), C.byref(kb))
    if e != 0:
        raise ValueError(f'bqio_inflate2: error {e}')
    return (oa.tobytes() if ka.value else None), (ob.tobytes() if kb.value else None)


def decode_jpeg(raw, tile_px=299):
    """One JPEG file's bytes -> uint8 [px,px,3] through the reader's own baseline decoder (csrc/jpeg_baseline.h);
    UnsupportedImage for streams outside its subset, ValueError for a tile of another size.  For tests."""
    out = np.empty((tile_px, tile_px, 3), np.uint8)
    e = lib().bqio_decode_jpeg(bytes(raw), len(raw), tile_px, out.ctypes.data)
    if e == ERR_UNSUPPORTED:
        raise UnsupportedImage(0)
    if e != 0:
        raise ValueError(f'bqio_decode_jpeg: error {e}')
    return out


def inflate_fallbacks():
    """Streams handed to zlib after the reader's own decompressor refused them although zlib accepts them (0 = none)."""
    return int(lib().bqio_inflate_fallbacks())


class UnsupportedImage(ValueError):
    def __init__(self, index):
        super().__init__(f'record {index}: image_raw is not a PNG or baseline JPEG the native decoder handles')
        self.index = index


def default_threads():
    """Decoder threads of one process: the cores it may use (its affinity mask, which ``distributed.pin_rank`` narrows to the
    rank's share of the node, capped by the cgroup's CPU quota), at most 64: a 64-core share decodes on 64 threads -- twice what
    one GPU consumes --, and eight ranks on one node do not start 8 x 16 threads on the same cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:    # cgroup v2 quota
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


class NativeReader:
    def __init__(self, path, verify='length'):
        self._lib = lib()
        self._h = self._lib.bqio_open(os.fsencode(path), VERIFY[verify])
        if not self._h:
            raise IOError(self._lib.bqio_last_error(None).decode())
        self.path = path

    def close(self):
        if self._h:
            self._lib.bqio_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()

    def __len__(self):
        return int(self._lib.bqio_count(self._h))

    @property
    def slide(self):
        buf = C.create_string_buffer(512)
        n = self._lib.bqio_slide_name(self._h, buf, 512)
        if n < 0:
            raise IOError(self._lib.bqio_last_error(self._h).decode())
        return buf.value.decode() if n else None

    def image_format(self, index):
        return int(self._lib.bqio_image_format(self._h, index))

    def image_bytes(self, index):
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        if self._lib.bqio_image_bytes(self._h, index, C.byref(p), C.byref(n)) != 0:
            raise IOError(self._lib.bqio_last_error(self._h).decode())
        return C.string_at(p, n.value)

    def probe(self, tile_px=299, first=0, count=None):
        """None when ``decode`` would take records [first, first + count) as far as that shows without decoding (``bqio_probe``:
        record framing, image signatures, the markers and scan structure of JPEG records); otherwise the index of the first
        record it would refuse.  One pass over the bytes of the JPEG records, nothing for PNG records."""
        total = len(self)
        count = total - first if count is None else count
        bad = _i64(-1)
        e = self._lib.bqio_probe(self._h, first, count, tile_px, C.byref(bad))
        return None if e == 0 else int(bad.value)

    def extract_z(self, first, count, tile_px, out_z, off, length, threads=None):
        """The zlib streams of records [first, first + count) packed into ``out_z`` (uint8 array, usually pinned) for the device
        inflate (``Engine.png_inflate``): ``off`` / ``length`` (uint32 [count]) receive every stream's offset (a multiple of 16)
        and size.  Returns (bytes used, loc int64 [count, 2]).  ``UnsupportedImage`` for a record that is not an 8-bit RGB PNG
        tile of ``tile_px``; ``MemoryError`` (with the bytes needed in ``.args[1]``) when ``out_z`` is too small."""
        assert out_z.dtype == np.uint8 and out_z.flags['C_CONTIGUOUS'] and off.dtype == np.uint32 and length.dtype == np.uint32
        assert off.size >= count and length.size >= count
        loc = np.zeros((count, 2), np.int64)
        used = C.c_size_t(0)
        bad = _i64(-1)
        e = self._lib.bqio_extract_z(self._h, first, count, tile_px, out_z.ctypes.data, out_z.size, off.ctypes.data,
                                     length.ctypes.data, loc.ctypes.data, C.byref(used), threads or default_threads(), C.byref(bad))
        if e == ERR_UNSUPPORTED:
            raise UnsupportedImage(bad.value)
        if e == ERR_FORMAT:
            raise ValueError(f'{self.path}: record {bad.value}: tile size differs from {(tile_px, tile_px, 3)}')
        if e != 0 and used.value > out_z.size:
            raise MemoryError(f'extract_z: {used.value} bytes needed, {out_z.size} given', used.value)
        if e != 0:
            raise IOError(f'{self.path}: {self._lib.bqio_last_error(self._h).decode()} (record {bad.value})')
        return int(used.value), loc

    def decode(self, first=0, count=None, tile_px=299, out=None, threads=None, rows=False):
        """-> (tiles uint8 [count,px,px,3], loc int64 [count,2]).  `out`: optional C-contiguous uint8
        array / tensor-backed numpy view to decode into (e.g. pinned memory).
        rows=True: the PNG scanline filters stay in -- [count,px,1+3*px], filter-type byte + filtered bytes per row, for
        `Engine.png_unfilter` on the GPU (tiles that are not 8-bit RGB PNGs arrive decoded, as rows of filter type 0)."""
        total = len(self)
        count = total - first if count is None else count
        shape = (count, tile_px, 1 + 3 * tile_px) if rows else (count, tile_px, tile_px, 3)
        if out is None:
            out = np.empty(shape, np.uint8)
        assert out.dtype == np.uint8 and out.flags['C_CONTIGUOUS'] and out.size == int(np.prod(shape))
        loc = np.zeros((count, 2), np.int64)
        bad = _i64(-1)
        fn = self._lib.bqio_decode_rows if rows else self._lib.bqio_decode
        e = fn(self._h, first, count, tile_px, out.ctypes.data, loc.ctypes.data, threads or default_threads(), C.byref(bad))
        if e == ERR_UNSUPPORTED:
            raise UnsupportedImage(bad.value)
        if e == ERR_FORMAT:       # same exception the Python reader raises for a tile of the wrong size
            raise ValueError(f'{self.path}: record {bad.value}: tile size differs from {(tile_px, tile_px, 3)}')
        if e != 0:
            raise IOError(f'{self.path}: {self._lib.bqio_last_error(self._h).decode()} (record {bad.value})')
        return out, loc
