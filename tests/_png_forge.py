"""Test helper: a PNG encoder that applies the scanline filter type the caller names for every row (Pillow chooses its own),
so that every filter type and every row-to-row combination can be put in front of a decoder."""
import struct
import zlib

import numpy as np


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def filter_rows(img, types):
    """img uint8 [H,W,3], types [H] in 0..4 -> uint8 [H, 1 + 3W]: filter-type byte + filtered bytes per row."""
    h, w, _ = img.shape
    flat = img.reshape(h, 3 * w).astype(np.int64)
    out = np.zeros((h, 1 + 3 * w), np.uint8)
    for y in range(h):
        cur = flat[y]
        up = flat[y - 1] if y else np.zeros_like(cur)
        a = np.concatenate([np.zeros(3, np.int64), cur[:-3]])
        c = np.concatenate([np.zeros(3, np.int64), up[:-3]])
        ft = int(types[y])
        pred = [np.zeros_like(cur), a, up, (a + up) >> 1, _paeth(a, up, c)][ft]
        out[y, 0] = ft
        out[y, 1:] = (cur - pred) & 255
    return out


def encode_png(img, types, level=6):
    h, w, _ = img.shape
    raw = filter_rows(img, types).tobytes()

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, level)) + chunk(b'IEND', b''))
