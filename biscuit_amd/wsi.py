"""A slide reader under the heatmap front-end: what ``sf.WSI(slide, 299, 302, roi_method='ignore')`` and
``wsi.build_generator(shuffle=False, include_loc='grid')`` give the reference's Figure-5 code (``results.py:216-265``) -- the tile grid
of a whole-slide image at a tile width in MICRONS, each tile resampled to ``tile_px`` -- for pyramidal TIFF slides, without Slideflow,
libvips or OpenSlide (none of which exists here).

What is read: classic TIFF and BigTIFF, little or big endian; 8-bit RGB / YCbCr pages in chunky layout, stored as strips or tiles;
compression none (1), zlib / Adobe deflate (8, 32946; horizontal predictor 2 supported) and JPEG (7: abbreviated streams + the page's
``JPEGTables``, decoded by Pillow's libjpeg).  That covers Aperio SVS written with JPEG tiles and generic pyramidal TIFFs (``vips
tiffsave --pyramid``, Pillow); JPEG 2000 SVS (33003 / 33005), LZW and planar layouts are refused with a message, not mis-read.
Microns per pixel come from the Aperio description (``MPP = 0.2520``) or from XResolution / ResolutionUnit.

PARITY UNPINNED (like everything on the producer side): Slideflow's reader is not available, no real slide exists here, and the kernel
libvips uses to shrink a region to ``tile_px`` is not reproduced -- tiles are resampled with Pillow's LANCZOS.  What IS pinned
(tests/test_wsi.py): the container layer against libtiff (through Pillow) on files Pillow writes, and against hand-assembled tiled /
BigTIFF / JPEGTables files; the grid arithmetic against its definition.
"""
import io
import re
import struct
import zlib

import numpy as np

TILE_PX, TILE_UM = 299, 302             # biscuit/hp.py:5, results.py:235


class SlideError(ValueError):
    pass


def _guard(fn):
    """A damaged file must end in SlideError, whatever the parser tripped over."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except SlideError:
            raise
        except (struct.error, IndexError, KeyError, TypeError, ValueError, ZeroDivisionError, OverflowError, MemoryError, OSError) as e:
            raise SlideError(f'damaged or unsupported slide file: {type(e).__name__}: {e}') from None
    return wrapped


_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 6: ('b', 1), 7: ('B', 1), 8: ('h', 2), 9: ('i', 4),
          10: ('ii', 8), 11: ('f', 4), 12: ('d', 8), 13: ('I', 4), 16: ('Q', 8), 17: ('q', 8), 18: ('Q', 8)}


class _Page:
    """One IFD that holds an 8-bit RGB / YCbCr image."""
    def __init__(self, tags):
        g = tags.get
        self.width, self.height = int(g(256)[0]), int(g(257)[0])
        self.bits = tuple(g(258, (1,)))
        self.compression = int(g(259, (1,))[0])
        self.photometric = int(g(262, (2,))[0])
        self.samples = int(g(277, (1,))[0])
        self.planar = int(g(284, (1,))[0])
        self.predictor = int(g(317, (1,))[0])
        self.subfile = int(g(254, (0,))[0])
        d = g(270, b'')
        self.description = (d if isinstance(d, bytes) else b'').split(b'\0')[0].decode('latin-1')
        self.jpeg_tables = bytes(g(347)) if g(347) is not None else None
        self.subsampling = tuple(g(530, (2, 2)))
        self.xres, self.yres, self.res_unit = g(282), g(283), int(g(296, (2,))[0])
        if g(322) is not None:           # tiles
            self.tiled = True
            self.tw, self.th = int(g(322)[0]), int(g(323)[0])
            self.offsets, self.counts = g(324), g(325)
        else:
            self.tiled = False
            self.tw = self.width
            self.th = int(g(278, (self.height,))[0])
            self.th = min(self.th, self.height)
            self.offsets, self.counts = g(273), g(279)
        if not (0 < self.width <= 1 << 20 and 0 < self.height <= 1 << 20 and 0 < self.tw <= 1 << 15 and 0 < self.th <= 1 << 15):
            raise SlideError(f'implausible page geometry {self.width} x {self.height}, segments {self.tw} x {self.th}')
        self.across = -(-self.width // self.tw)
        self.down = -(-self.height // self.th)

    def usable(self):
        return self.samples == 3 and self.bits[:3] == (8, 8, 8) and self.offsets is not None and self.counts is not None

    def check(self):
        if self.planar != 1:
            raise SlideError('planar (separate-plane) TIFF pages are not supported')
        if self.compression in (33003, 33005, 34712):
            raise SlideError('JPEG 2000 compressed slide: not supported (no decoder in this image)')
        if self.compression == 5:
            raise SlideError('LZW-compressed TIFF: not supported (deflate, JPEG and uncompressed are)')
        if self.compression not in (1, 7, 8, 32946):
            raise SlideError(f'TIFF compression {self.compression} is not supported')
        if len(self.offsets) != len(self.counts) or len(self.offsets) < self.across * self.down:
            raise SlideError('TIFF page: strip / tile tables do not cover the image')


class TiffSlide:
    """``TiffSlide(path)``: ``levels`` (pyramid pages, largest first), ``level_downsamples``, ``dimensions`` (width, height of level 0),
    ``mpp`` (microns per pixel of level 0, or None), ``read_region(level, x, y, w, h)`` in that level's pixels -> uint8 [h, w, 3]
    (white outside the image, as slide viewers pad)."""

    def __init__(self, path):
        self.path = path
        self._f = open(path, 'rb')
        try:
            self._open()
        except BaseException:
            self._f.close()
            raise

    @_guard
    def _open(self):
        import os
        self._size = os.fstat(self._f.fileno()).st_size
        path = self.path
        head = self._f.read(16)
        if head[:2] == b'II':
            self._e = '<'
        elif head[:2] == b'MM':
            self._e = '>'
        else:
            raise SlideError(f'{path}: not a TIFF file')
        magic = struct.unpack(self._e + 'H', head[2:4])[0]
        if magic == 42:
            self._big, first = False, struct.unpack(self._e + 'I', head[4:8])[0]
        elif magic == 43:
            if struct.unpack(self._e + 'HH', head[4:8]) != (8, 0):
                raise SlideError(f'{path}: malformed BigTIFF header')
            self._big, first = True, struct.unpack(self._e + 'Q', head[8:16])[0]
        else:
            raise SlideError(f'{path}: not a TIFF file (magic {magic})')
        pages, off, seen = [], first, set()
        while off and off not in seen and len(pages) < 64:
            seen.add(off)
            tags, off = self._ifd(off)
            if 256 in tags and 257 in tags:
                p = _Page(tags)
                if p.usable():
                    pages.append(p)
        if not pages:
            raise SlideError(f'{path}: no 8-bit RGB page')
        # the pyramid: the largest page and every smaller page of the same aspect ratio that is not a label / macro / thumbnail strip
        # image of an Aperio file (those are stripped and named in their descriptions; a pyramid level of an SVS is tiled)
        base = max(pages, key=lambda p: p.width * p.height)
        levels = [base]
        for p in sorted(pages, key=lambda p: -p.width):
            if p is base or p.width >= levels[-1].width:
                continue
            same = abs(p.width / base.width - p.height / base.height) < 0.02
            named = re.search(r'\b(label|macro)\b', p.description, re.I) is not None
            if same and not named and (p.tiled or not base.tiled):
                levels.append(p)
        for p in levels:
            p.check()
        self.levels = levels
        self.dimensions = (base.width, base.height)
        self.level_dimensions = [(p.width, p.height) for p in levels]
        self.level_downsamples = [base.width / p.width for p in levels]
        self.mpp = self._mpp(base)
        self._cache = {}

    # ---- container ---------------------------------------------------------------------------------------------------
    def _ifd(self, off):
        f, e = self._f, self._e
        f.seek(off)
        if self._big:
            n = struct.unpack(e + 'Q', f.read(8))[0]
            raw = f.read(20 * n + 8)
            esz, fmt, inl = 20, 'HHQ', 8
        else:
            n = struct.unpack(e + 'H', f.read(2))[0]
            raw = f.read(12 * n + 4)
            esz, fmt, inl = 12, 'HHI', 4
        if n > 4096 or len(raw) < esz * n:
            raise SlideError(f'{self.path}: corrupt IFD')
        tags = {}
        for i in range(n):
            ent = raw[i * esz:(i + 1) * esz]
            tag, typ, cnt = struct.unpack(e + fmt, ent[:esz - inl])
            if typ not in _TYPES:
                continue
            code, size = _TYPES[typ]
            nbytes = size * cnt
            if nbytes > self._size:
                raise SlideError(f'{self.path}: tag {tag} claims {nbytes} bytes in a file of {self._size}')
            if nbytes <= inl:
                data = ent[esz - inl:esz - inl + nbytes]
            else:
                pos = struct.unpack(e + ('Q' if self._big else 'I'), ent[esz - inl:])[0]
                f.seek(pos)
                data = f.read(nbytes)
                if len(data) != nbytes:
                    raise SlideError(f'{self.path}: tag {tag} runs past the end of the file')
            if typ in (2, 7) or (typ == 1 and tag in (270, 347)):
                tags[tag] = bytes(data)
            elif typ in (5, 10):
                v = struct.unpack(e + code[0] * (2 * cnt), data)
                tags[tag] = tuple((v[2 * k], v[2 * k + 1]) for k in range(cnt))
            else:
                tags[tag] = struct.unpack(e + code * cnt, data)
        nxt = struct.unpack(e + ('Q' if self._big else 'I'), raw[esz * n:esz * n + (8 if self._big else 4)])[0]
        return tags, nxt

    @staticmethod
    def _mpp(p):
        m = re.search(r'MPP\s*=\s*([0-9.]+)', p.description)
        if m:
            return float(m.group(1))
        if p.xres and p.xres[0][1] and p.xres[0][0] and p.res_unit in (2, 3):
            per_unit = p.xres[0][0] / p.xres[0][1]
            return (25400.0 if p.res_unit == 2 else 10000.0) / per_unit
        return None

    # ---- pixels ------------------------------------------------------------------------------------------------------
    @_guard
    def _segment(self, li, index):
        """Strip / tile ``index`` of level ``li`` decoded to uint8 [th, tw, 3] (a last strip may be shorter)."""
        key = (li, index)
        if key in self._cache:
            return self._cache[key]
        p = self.levels[li]
        if int(p.counts[index]) > self._size:
            raise SlideError(f'{self.path}: segment {index} of level {li} claims more bytes than the file holds')
        self._f.seek(int(p.offsets[index]))
        raw = self._f.read(int(p.counts[index]))
        rows = p.th if p.tiled else min(p.th, p.height - (index * p.th))
        if p.compression == 7:
            img = self._jpeg(p, raw)
            if img.shape[0] < rows or img.shape[1] < p.tw:
                raise SlideError(f'{self.path}: JPEG segment {index} of level {li} is smaller than its tile')
            img = img[:rows, :p.tw]
        else:
            if p.compression in (8, 32946):
                try:
                    raw = zlib.decompress(raw)
                except zlib.error as e:
                    raise SlideError(f'{self.path}: damaged deflate segment {index} of level {li}: {e}') from None
            need = rows * p.tw * 3
            if len(raw) < need:
                raise SlideError(f'{self.path}: segment {index} of level {li} is short ({len(raw)} of {need} bytes)')
            img = np.frombuffer(raw, np.uint8, need).reshape(rows, p.tw, 3)
            if p.predictor == 2:
                img = np.cumsum(img, axis=1, dtype=np.uint8)          # horizontal differencing, per sample, modulo 256
            elif p.predictor != 1:
                raise SlideError(f'TIFF predictor {p.predictor} is not supported')
            if p.photometric == 6:
                raise SlideError('uncompressed YCbCr TIFF pages are not supported')
        if len(self._cache) > 64:
            self._cache.pop(next(iter(self._cache)))
        self._cache[key] = img
        return img

    def _jpeg(self, p, raw):
        from PIL import Image
        if p.jpeg_tables:
            t = p.jpeg_tables
            t = t[:-2] if t.endswith(b'\xff\xd9') else t
            raw = t + (raw[2:] if raw[:2] == b'\xff\xd8' else raw)
        try:
            im = Image.open(io.BytesIO(raw))
            if p.photometric == 2 and im.mode == 'YCbCr':
                raise SlideError('RGB-photometric JPEG tiles whose streams carry no colour-space marker are not supported')
            return np.asarray(im.convert('RGB'))
        except (OSError, SyntaxError) as e:
            raise SlideError(f'{self.path}: a JPEG tile does not decode: {e}') from None

    @_guard
    def read_region(self, level, x, y, w, h):
        p = self.levels[level]
        if not (0 < w <= 1 << 16 and 0 < h <= 1 << 16):
            raise SlideError(f'region of {w} x {h} pixels')
        out = np.full((h, w, 3), 255, np.uint8)
        x0, y0, x1, y1 = max(x, 0), max(y, 0), min(x + w, p.width), min(y + h, p.height)
        if x1 <= x0 or y1 <= y0:
            return out
        for ty in range(y0 // p.th, (y1 - 1) // p.th + 1):
            for tx in range(x0 // p.tw, (x1 - 1) // p.tw + 1):
                seg = self._segment(level, ty * p.across + tx)
                sx0, sy0 = tx * p.tw, ty * p.th
                ax0, ay0 = max(x0, sx0), max(y0, sy0)
                ax1, ay1 = min(x1, sx0 + seg.shape[1]), min(y1, sy0 + seg.shape[0])
                if ax1 > ax0 and ay1 > ay0:
                    out[ay0 - y:ay1 - y, ax0 - x:ax1 - x] = seg[ay0 - sy0:ay1 - sy0, ax0 - sx0:ax1 - sx0]
        return out

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class WSI:
    """The tile grid of one slide at ``tile_um`` microns per tile, each tile ``tile_px`` pixels wide: the object the reference builds
    with ``sf.WSI(slide, 299, 302, roi_method='ignore')`` (results.py:235).  ``extract_px`` = the tile's width in level-0 pixels
    (``tile_um / mpp``), the grid walks the slide at stride ``extract_px / stride_div`` (border remainders dropped), every tile is read
    from the pyramid level with the largest downsample that still has at least ``tile_px`` pixels per tile and resampled to
    ``tile_px`` (Pillow LANCZOS).  ``roi_method``: only 'ignore' (what the reference passes) is implemented."""

    @_guard
    def __init__(self, path, tile_px=TILE_PX, tile_um=TILE_UM, stride_div=1, roi_method='ignore', mpp=None):
        if roi_method != 'ignore':
            raise NotImplementedError("only roi_method='ignore' (results.py:235)")
        self.slide = TiffSlide(path)
        self.path, self.tile_px, self.tile_um, self.stride_div = path, int(tile_px), float(tile_um), int(stride_div)
        self.mpp = float(mpp) if mpp else self.slide.mpp
        if not self.mpp:
            raise SlideError(f'{path}: no microns-per-pixel in the file (Aperio "MPP =" or XResolution); pass mpp=')
        self.extract_px = int(self.tile_um / self.mpp)                   # level-0 pixels per tile side
        if not np.isfinite(self.mpp) or self.mpp <= 0:
            raise SlideError(f'{path}: implausible microns per pixel {self.mpp}')
        if self.extract_px < 1 or self.stride_div < 1:
            raise SlideError('tile of less than one pixel / bad stride_div')
        self.stride = max(1, self.extract_px // self.stride_div)
        w, h = self.slide.dimensions
        self.grid_w = (w - self.extract_px) // self.stride + 1 if w >= self.extract_px else 0
        self.grid_h = (h - self.extract_px) // self.stride + 1 if h >= self.extract_px else 0
        want = self.extract_px / self.tile_px                            # the downsample the tile asks for
        ok = [i for i, d in enumerate(self.slide.level_downsamples) if d <= want * 1.0001]
        self.level = max(ok, key=lambda i: self.slide.level_downsamples[i]) if ok else 0
        self.level_ds = self.slide.level_downsamples[self.level]
        self.estimated_num_tiles = self.grid_w * self.grid_h

    def _tile(self, gx, gy):
        from PIL import Image
        x0, y0 = gx * self.stride, gy * self.stride
        lx, ly = int(round(x0 / self.level_ds)), int(round(y0 / self.level_ds))
        lw = max(1, int(round(self.extract_px / self.level_ds)))
        reg = self.slide.read_region(self.level, lx, ly, lw, lw)
        if lw != self.tile_px:
            reg = np.asarray(Image.fromarray(reg).resize((self.tile_px, self.tile_px), Image.LANCZOS))
        return reg

    def build_generator(self, shuffle=False, include_loc='grid', show_progress=False, **_):
        """-> a zero-argument callable whose generator yields ``{'image': uint8 [tile_px, tile_px, 3], 'loc' (and 'grid'): (gx, gy)}``
        in row-major grid order -- the dictionaries results.py:236-262 consumes under either Slideflow version's key."""
        if shuffle:
            raise NotImplementedError('shuffle=False is what the reference asks for (results.py:237)')

        def gen():
            for gy in range(self.grid_h):
                for gx in range(self.grid_w):
                    yield {'image': self._tile(gx, gy), 'loc': (gx, gy), 'grid': (gx, gy)}
        return gen

    def tiles(self):
        """(tiles uint8 [T, tile_px, tile_px, 3], grid int64 [T, 2] of (gx, gy)) -- everything at once, for ``heatmap.Heatmap``."""
        n = self.grid_w * self.grid_h
        out = np.empty((n, self.tile_px, self.tile_px, 3), np.uint8)
        grid = np.empty((n, 2), np.int64)
        for i, t in enumerate(self.build_generator()()):
            out[i] = t['image']
            grid[i] = t['loc']
        return out, grid

    def close(self):
        self.slide.close()
