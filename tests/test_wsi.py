"""The slide reader under the heatmap front-end (``biscuit_amd/wsi.py``; SURVEY.md section 8f row 4, reference: ``results.py:216-265``).
CPU: the TIFF container layer against libtiff (through Pillow) on files Pillow writes -- strips; none / deflate / JPEG; a multi-page
pyramid with resolution tags -- and against hand-assembled files for what Pillow does not write: TILES, BigTIFF, big-endian, the
horizontal predictor, abbreviated JPEG tiles + JPEGTables, an Aperio description.  Then the tile grid ``sf.WSI(slide, 299, 302)`` defines.
GPU: ``Heatmap.from_slide`` = ``Heatmap`` on the reader's tiles.

PARITY UNPINNED against Slideflow (not installable; no real slide here): see the module's header."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from biscuit_amd.wsi import WSI, SlideError, TiffSlide


def _img(w, h, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / 37 + seed), 128 + 80 * np.cos(yy / 29), 128 + 60 * np.sin((xx + yy) / 53)], -1)
    return np.clip(base + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize('comp', ['raw', 'tiff_adobe_deflate', 'jpeg'])
def test_pillow_written_pyramid_reads_like_libtiff(tmp_path, comp):
    a = _img(1000, 700, 1)
    levels = [Image.fromarray(a), Image.fromarray(a).resize((500, 350), Image.BILINEAR), Image.fromarray(a).resize((250, 175), Image.BILINEAR)]
    path = str(tmp_path / 's.tif')
    levels[0].save(path, format='TIFF', compression=comp, save_all=True, append_images=levels[1:], dpi=(25400 / 0.5, 25400 / 0.5))
    s = TiffSlide(path)
    assert s.dimensions == (1000, 700) and s.level_dimensions == [(1000, 700), (500, 350), (250, 175)]
    assert np.allclose(s.level_downsamples, [1, 2, 4]) and abs(s.mpp - 0.5) < 1e-6
    ref = Image.open(path)
    for li in range(3):
        ref.seek(li)
        want = np.asarray(ref.convert('RGB'))
        w, h = s.level_dimensions[li]
        got = s.read_region(li, 0, 0, w, h)
        if comp == 'jpeg':            # both sides decode the same strips with libjpeg; colour conversion / upsampling settings may differ
            d = np.abs(got.astype(int) - want.astype(int))
            assert d.max() <= 3 and (d > 1).mean() < 0.02, (li, d.max(), (d > 1).mean())
        else:
            assert np.array_equal(got, want), li
        # a window across strip borders, partly outside the image: white outside
        win = s.read_region(li, w - 40, h - 30, 100, 80)
        assert np.array_equal(win[:30, :40], got[h - 30:, w - 40:]) and (win[30:] == 255).all() and (win[:, 40:] == 255).all()
    s.close()


def _tiff(pages, big=False, endian='<'):
    """Hand-assembled TIFF: pages = [dict(w, h, tw, th, segs [bytes], comp, photometric, predictor, desc, tables, tiled)]."""
    e = endian
    out = bytearray()
    out += (b'II' if e == '<' else b'MM') + (struct.pack(e + 'HHHQ', 43, 8, 0, 0) if big else struct.pack(e + 'HI', 42, 0))
    fix_first = len(out) - (8 if big else 4)
    ifd_ptr_at = fix_first
    for pg in pages:
        seg_off = []
        for sg in pg['segs']:
            seg_off.append(len(out)); out += sg
            if len(out) & 1:
                out += b'\0'
        ents = []           # (tag, type, count, value bytes)

        def add(tag, typ, vals):
            code = {3: 'H', 4: 'I', 16: 'Q'}.get(typ)
            if typ == 2:
                data = vals + b'\0'; cnt = len(data)
            elif typ == 7:
                data = vals; cnt = len(data)
            else:
                data = struct.pack(e + code * len(vals), *vals); cnt = len(vals)
            ents.append((tag, typ, cnt, data))
        lt = 16 if big else 4
        add(256, 4, [pg['w']]); add(257, 4, [pg['h']]); add(258, 3, [8, 8, 8]); add(259, 3, [pg['comp']])
        add(262, 3, [pg.get('photometric', 2)])
        if pg.get('desc'):
            add(270, 2, pg['desc'].encode())
        add(277, 3, [3]); add(284, 3, [1])
        if pg.get('predictor', 1) != 1:
            add(317, 3, [pg['predictor']])
        if pg.get('tiled', True):
            add(322, 4, [pg['tw']]); add(323, 4, [pg['th']]); add(324, lt, seg_off); add(325, lt, [len(x) for x in pg['segs']])
        else:
            add(278, 4, [pg['th']]); add(273, lt, seg_off); add(279, lt, [len(x) for x in pg['segs']])
        if pg.get('tables'):
            add(347, 7, pg['tables'])
        ents.sort(key=lambda t: t[0])
        inl = 8 if big else 4
        big_data = []
        for (tag, typ, cnt, data) in ents:
            if len(data) > inl:
                big_data.append((tag, len(out))); out += data
                if len(out) & 1:
                    out += b'\0'
        where = dict(big_data)
        ifd_at = len(out)
        out[ifd_ptr_at:ifd_ptr_at + (8 if big else 4)] = struct.pack(e + ('Q' if big else 'I'), ifd_at)
        out += struct.pack(e + ('Q' if big else 'H'), len(ents))
        for (tag, typ, cnt, data) in ents:
            out += struct.pack(e + ('HHQ' if big else 'HHI'), tag, typ, cnt)
            out += struct.pack(e + ('Q' if big else 'I'), where[tag]) if len(data) > inl else data + b'\0' * (inl - len(data))
        ifd_ptr_at = len(out)
        out += struct.pack(e + ('Q' if big else 'I'), 0)
    return bytes(out)


def _tiles_of(a, tw, th, enc):
    h, w = a.shape[:2]
    segs = []
    for ty in range(-(-h // th)):
        for tx in range(-(-w // tw)):
            t = np.full((th, tw, 3), 7, np.uint8)                  # (padding of border tiles: any value, never shown)
            blk = a[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            t[:blk.shape[0], :blk.shape[1]] = blk
            segs.append(enc(t))
    return segs


@pytest.mark.parametrize('big,endian', [(False, '<'), (True, '<'), (False, '>'), (True, '>')])
def test_tiled_pages_bigtiff_endianness_and_predictor(tmp_path, big, endian):
    a = _img(600, 420, 2)
    b = np.asarray(Image.fromarray(a).resize((300, 210), Image.BILINEAR))

    def deflate_pred(t):
        d = t.astype(np.int16)
        d[:, 1:] -= t[:, :-1].astype(np.int16)
        return zlib.compress((d & 255).astype(np.uint8).tobytes())
    pages = [dict(w=600, h=420, tw=256, th=128, comp=8, predictor=2, segs=_tiles_of(a, 256, 128, deflate_pred), desc='Aperio Image Library v1\n600x420 |AppMag = 20|MPP = 0.4990'),
             dict(w=300, h=210, tw=128, th=128, comp=1, segs=_tiles_of(b, 128, 128, lambda t: t.tobytes())),
             dict(w=150, h=52, tw=150, th=52, comp=1, tiled=False, segs=[np.zeros((52, 150, 3), np.uint8).tobytes()], desc='label 150x52')]
    path = tmp_path / 't.svs'
    path.write_bytes(_tiff(pages, big=big, endian=endian))
    s = TiffSlide(str(path))
    assert s.level_dimensions == [(600, 420), (300, 210)]                   # the label strip image is not a pyramid level
    assert abs(s.mpp - 0.499) < 1e-9
    assert np.array_equal(s.read_region(0, 0, 0, 600, 420), a) and np.array_equal(s.read_region(1, 0, 0, 300, 210), b)
    assert np.array_equal(s.read_region(0, 250, 120, 20, 20), a[120:140, 250:270])      # across four tiles
    assert (s.read_region(0, -10, -10, 20, 20)[:10] == 255).all() and np.array_equal(s.read_region(0, -10, -10, 20, 20)[10:, 10:], a[:10, :10])


def test_jpeg_tiles_with_shared_tables(tmp_path):
    a = _img(512, 512, 3)
    probe = io.BytesIO()
    Image.fromarray(a[:256, :256]).save(probe, format='JPEG', quality=85, streamtype=1)        # tables only
    tables = probe.getvalue()

    def enc(t):
        bb = io.BytesIO()
        Image.fromarray(t).save(bb, format='JPEG', quality=85, streamtype=2)                    # image data without tables
        return bb.getvalue()
    segs = _tiles_of(a, 256, 256, enc)
    assert all(b'\xff\xdb' not in sg[:64] for sg in segs) and b'\xff\xdb' in tables          # really abbreviated streams
    path = tmp_path / 'j.svs'
    path.write_bytes(_tiff([dict(w=512, h=512, tw=256, th=256, comp=7, photometric=6, segs=segs, tables=tables, desc='Aperio |MPP = 0.25')]))
    s = TiffSlide(str(path))
    got = s.read_region(0, 0, 0, 512, 512)
    for k, (ty, tx) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        bb = io.BytesIO()
        Image.fromarray(a[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256]).save(bb, format='JPEG', quality=85)
        want = np.asarray(Image.open(io.BytesIO(bb.getvalue())).convert('RGB'))              # the same tile as a complete stream
        assert np.array_equal(got[ty * 256:(ty + 1) * 256, tx * 256:(tx + 1) * 256], want), k
    assert np.abs(got.astype(int) - a.astype(int)).mean() < 8                                   # (and it is the picture)


def test_refusals(tmp_path):
    a = _img(64, 64, 4)
    for name, pg, msg in [('j2k', dict(w=64, h=64, tw=64, th=64, comp=33003, segs=[b'x' * 10]), 'JPEG 2000'),
                          ('lzw', dict(w=64, h=64, tw=64, th=64, comp=5, segs=[b'x' * 10]), 'LZW'),
                          ('short', dict(w=64, h=64, tw=64, th=64, comp=1, segs=[a.tobytes()[:100]]), None)]:
        p = tmp_path / f'{name}.tif'
        p.write_bytes(_tiff([pg]))
        if msg:
            with pytest.raises(SlideError, match=msg):
                TiffSlide(str(p))
        else:
            with pytest.raises(SlideError, match='short'):
                TiffSlide(str(p)).read_region(0, 0, 0, 64, 64)
    (tmp_path / 'x.tif').write_bytes(b'not a tiff at all')
    with pytest.raises(SlideError):
        TiffSlide(str(tmp_path / 'x.tif'))
    p = tmp_path / 'nompp.tif'
    p.write_bytes(_tiff([dict(w=64, h=64, tw=64, th=64, comp=1, segs=[a.tobytes()])]))
    with pytest.raises(SlideError, match='microns'):
        WSI(str(p))
    assert WSI(str(p), mpp=0.5, tile_px=16, tile_um=16).estimated_num_tiles == 4


def _slide_file(tmp_path, w=2400, h=1800, mpp=0.5045):
    a = _img(w, h, 5)
    b = np.asarray(Image.fromarray(a).resize((w // 4, h // 4), Image.BILINEAR))
    raw = lambda t: zlib.compress(t.tobytes(), 1)                                            # noqa: E731
    path = tmp_path / 'slide.svs'
    path.write_bytes(_tiff([dict(w=w, h=h, tw=256, th=256, comp=8, segs=_tiles_of(a, 256, 256, raw), desc=f'Aperio |MPP = {mpp}'),
                            dict(w=w // 4, h=h // 4, tw=256, th=256, comp=8, segs=_tiles_of(b, 256, 256, raw))]))
    return str(path), a


def test_wsi_grid_is_the_one_the_reference_asks_for(tmp_path):
    """``sf.WSI(slide, 299, 302)``: a tile is 302 um wide = int(302 / mpp) level-0 pixels, the grid walks the slide at that stride
    (stride_div = 1), tiles come back 299 x 299 in row-major order with their grid location under 'loc' and 'grid'."""
    path, a = _slide_file(tmp_path)
    w = WSI(path, 299, 302)
    assert w.extract_px == int(302 / 0.5045) == 598 and w.stride == 598
    assert (w.grid_w, w.grid_h) == ((2400 - 598) // 598 + 1, (1800 - 598) // 598 + 1) == (4, 3)
    assert w.level == 0 and w.level_ds == 1.0                     # 598 / 299 = 2.0: the 4 x level would have too few pixels
    items = list(w.build_generator(shuffle=False, include_loc='grid')())
    assert len(items) == 12 and [t['loc'] for t in items[:5]] == [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1)] and items[7]['grid'] == (3, 1)
    assert all(t['image'].shape == (299, 299, 3) and t['image'].dtype == np.uint8 for t in items)
    want = np.asarray(Image.fromarray(a[598:1196, 1196:1794]).resize((299, 299), Image.LANCZOS))
    assert np.array_equal(items[4 + 2]['image'], want)
    tiles, grid = w.tiles()
    assert tiles.shape == (12, 299, 299, 3) and grid.tolist()[6] == [2, 1] and np.array_equal(tiles[6], want)
    # a coarser request reads the second pyramid level: 800 um tiles = 1 585 px = 5.3 x 299 -> the 4 x level
    big = WSI(path, 299, 800)
    assert big.level == 1 and big.level_ds == 4.0 and big.estimated_num_tiles == 1
    # stride_div = 2: the overlapping grid of sf.Heatmap(..., stride_div=2)
    half = WSI(path, 299, 302, stride_div=2)
    assert half.stride == 299 and (half.grid_w, half.grid_h) == (7, 5)
    with pytest.raises(NotImplementedError):
        WSI(path, roi_method='inside')


@pytest.mark.gpu
def test_heatmap_from_slide_file(tmp_path):
    from biscuit_amd.engine import Engine
    from biscuit_amd.heatmap import Heatmap
    from biscuit_amd.weights import synthetic_weights
    path, _ = _slide_file(tmp_path)
    eng = Engine(synthetic_weights(1), dtype='f16', max_batch=16, max_mc=8)
    hm = Heatmap.from_slide(eng, path, mc_n=8, seed=3, batch=16)
    tiles, grid = WSI(path).tiles()
    ref = Heatmap(eng, tiles, grid, grid_shape=(3, 4), mc_n=8, seed=3, batch=16)
    assert hm.logits.shape == (3, 4, 2) and np.array_equal(hm.logits, ref.logits) and np.array_equal(hm.uncertainty, ref.uncertainty)
    assert (hm.uncertainty[:, :, 0] > 0).all() and np.allclose(hm.logits.sum(-1), 1.0, atol=1e-5)
    incl, excl = hm.split_by_uncertainty(float(np.median(hm.uncertainty[:, :, 0])))
    assert len(incl) + len(excl) == 12
    eng.close()


def test_mutated_slide_files_fail_with_slide_errors_only(tmp_path):
    """Bytes flipped anywhere in a small slide file: the reader answers with pixels or with ``SlideError`` -- never with another
    exception, a hang or a huge allocation."""
    path, _ = _slide_file(tmp_path, w=600, h=420)
    good = bytearray(open(path, 'rb').read())
    rng = np.random.default_rng(0)
    # the structure lives at the end (IFDs, tag data): mutate there densely and the pixel data sparsely
    spots = np.concatenate([rng.integers(0, 16, 40), rng.integers(len(good) - 1200, len(good), 500), rng.integers(0, len(good), 60)])
    outcomes = {'ok': 0, 'refused': 0}
    for k, at in enumerate(spots):
        bad = bytearray(good)
        bad[int(at)] ^= int(rng.integers(1, 256))
        if k % 5 == 0:
            bad = bad[:int(at)]                                   # truncation
        p = tmp_path / 'm.svs'
        p.write_bytes(bytes(bad))
        try:
            w = WSI(str(p), 299, 302)
            if w.estimated_num_tiles:
                next(iter(w.build_generator()()))
            w.close()
            outcomes['ok'] += 1
        except SlideError:
            outcomes['refused'] += 1
    assert outcomes['ok'] > 20 and outcomes['refused'] > 20, outcomes
