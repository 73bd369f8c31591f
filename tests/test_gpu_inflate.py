"""The device inflate (csrc/kernels_inflate.hip, ``bq_png_inflate``) against zlib, byte for byte: ``-m gpu``.

zlib is the oracle here -- the reference's tiles are decoded by ``tf.io.decode_png`` / libpng, i.e. by zlib's inflate; a stream
zlib's ``decompress`` accepts must give the same bytes on the device, a stream it refuses must be flagged (status != 0)."""
import os
import zlib

import numpy as np
import pytest
import torch

from biscuit_amd import tfrecord as tfr
from biscuit_amd import tfrecord_native as tn
from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', params=[0, 5], ids=['tables_in_l2', 'rounds_7bit'])
def eng(request):
    from biscuit_amd.engine import Engine
    e = Engine(synthetic_weights(1), dtype='f16', max_batch=8, max_mc=2)
    e.set_option('inflate_variant', request.param)
    return e


def pack(streams):
    """zlib streams -> (z uint8, off int32, len int32) in the layout of bqio_extract_z: 16-byte aligned starts, >= 32 zero bytes behind."""
    off, chunks, at = [], [], 0
    for s in streams:
        off.append(at)
        pad = (-(len(s) + 32)) % 16 + 32
        chunks.append(bytes(s) + b'\0' * pad)
        at += len(s) + pad
    return (np.frombuffer(b''.join(chunks), np.uint8).copy(), np.array(off, np.int32), np.array([len(s) for s in streams], np.int32))


def run(eng, streams, px):
    z, off, ln = pack(streams)
    rows, status = eng.png_inflate(torch.from_numpy(z).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(ln).cuda(), px)
    return rows.cpu().numpy()[:, :px * (1 + 3 * px)], status.cpu().numpy()


def payloads(px, rng):
    n = px * (1 + 3 * px)
    noise = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    narrow = np.clip(rng.normal(128, 6, n), 0, 255).astype(np.uint8).tobytes()          # short codes, some matches
    period = (bytes(range(7)) * (n // 7 + 1))[:n]                                        # distance 7, maximal lengths
    run1 = b'\x05' * n                                                                    # distance 1 (overlapping copies)
    text = (b'the quick brown fox jumps over the lazy dog. ' * (n // 45 + 1))[:n]
    skew = rng.choice(np.arange(256, dtype=np.uint8), n, p=np.r_[[0.9], np.full(255, 0.1 / 255)]).tobytes()   # one 1-bit code, many 12+ bit codes

    def scanlines(raw):          # every row starts with a PNG filter type (0..4): bq_png_inflate flags anything else
        a = np.frombuffer(raw, np.uint8).copy()
        a[::1 + 3 * px] %= 5
        return a.tobytes()
    return {k: scanlines(v) for k, v in {'noise': noise, 'narrow': narrow, 'period7': period, 'run': run1, 'text': text, 'skew': skew}.items()}


@pytest.mark.parametrize('px', [8, 64, 299])
def test_streams_of_every_block_kind_equal_zlib(eng, px):
    rng = np.random.default_rng(px)
    data = payloads(px, rng)
    streams, want, names = [], [], []
    for name, raw in data.items():
        for level in (0, 1, 6, 9):                                                       # 0: stored blocks
            streams.append(zlib.compress(raw, level)); want.append(raw); names.append(f'{name}/level{level}')
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)                      # fixed Huffman blocks
        streams.append(c.compress(raw) + c.flush()); want.append(raw); names.append(f'{name}/fixed')
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 1)                                    # memLevel 1: a block every 127 symbols
        streams.append(c.compress(raw) + c.flush()); want.append(raw); names.append(f'{name}/mem1')
        c = zlib.compressobj(6, zlib.DEFLATED, 9)                                        # a 512-byte window
        streams.append(c.compress(raw) + c.flush()); want.append(raw); names.append(f'{name}/win9')
        c = zlib.compressobj(6)                                                          # sync flushes: empty stored blocks in between
        third = len(raw) // 3
        streams.append(c.compress(raw[:third]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(raw[third:]) + c.flush())
        want.append(raw); names.append(f'{name}/syncflush')
    got, status = run(eng, streams, px)
    for i, nm in enumerate(names):
        assert status[i] == 0, (nm, int(status[i]))
        assert got[i].tobytes() == want[i], nm


def test_what_zlib_refuses_is_flagged(eng):
    px = 64
    n = px * (1 + 3 * px)
    rng = np.random.default_rng(1)
    arr = np.clip(rng.normal(128, 20, n), 0, 255).astype(np.uint8)
    arr[::1 + 3 * px] %= 5                                                               # rows start with a filter type 0..4
    raw = arr.tobytes()
    good = zlib.compress(raw, 6)
    cases = {'good': good}
    bad_ft = arr.copy(); bad_ft[17 * (1 + 3 * px)] = 5                                   # a well-formed stream whose row 17 has filter type 5:
    cases['filter_type'] = zlib.compress(bad_ft.tobytes(), 6)                            # zlib is content, no PNG decoder is
    cases['adler'] = good[:-1] + bytes([good[-1] ^ 1])                                   # trailer off by one bit
    cases['truncated'] = good[: len(good) // 2]
    cases['trailing'] = good + b'\x00\x01'
    cases['short_output'] = zlib.compress(raw[:-5], 6)
    cases['long_output'] = zlib.compress(raw + b'xyz', 6)
    cases['header'] = b'\x79' + good[1:]
    cases['dict'] = bytes([good[0], good[1] | 0x20]) + good[2:]
    cases['empty'] = b''
    for k in range(20):                                                                  # a flipped bit somewhere in the data
        pos = int(rng.integers(2, len(good) - 4))
        b = bytearray(good); b[pos] ^= 1 << int(rng.integers(0, 8))
        cases[f'flip{k}'] = bytes(b)
    names = list(cases)
    got, status = run(eng, [cases[k] for k in names], px)
    for i, nm in enumerate(names):
        try:
            ref = zlib.decompress(cases[nm])
            ok = len(ref) == n
        except zlib.error:
            ref, ok = None, False
        if nm in ('trailing', 'filter_type'):                                            # zlib.decompress tolerates trailing bytes
            ok = False                                                                   # (uncompress()-style strictness here)
        assert (status[i] == 0) == ok, (nm, int(status[i]), ok)
        if ok:
            assert got[i].tobytes() == ref, nm


def test_png_tiles_from_a_tfrecord_equal_the_host_decoder(eng, tmp_path):
    """TFRecord -> bqio_extract_z -> device inflate + un-filter = the host decoder's tiles, for nearly incompressible tiles,
    photo-like ones and flat ones, written at three compression levels."""
    import io
    from PIL import Image
    tiles = np.concatenate([make_tiles(5, seed=3), make_tiles(5, seed=4, grain=4.0), make_tiles(3, seed=5, grain=0.3),
                            np.full((1, 299, 299, 3), 77, np.uint8)])
    raws = []
    for i, t in enumerate(tiles):
        b = io.BytesIO()
        Image.fromarray(t).save(b, format='PNG', compress_level=(1, 6, 9)[i % 3])
        raws.append(b.getvalue())
    path = str(tmp_path / 's.tfrecords')
    locs = np.arange(2 * len(raws), dtype=np.int64).reshape(-1, 2)
    tfr.write_slide(path, 's', raws, locs)
    n = len(raws)
    with tn.NativeReader(path) as r:
        want, wloc = r.decode()
        z = np.zeros(n * 300000, np.uint8)
        off, ln = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        used, loc = r.extract_z(0, n, 299, z, off, ln)
        with pytest.raises(MemoryError):
            r.extract_z(0, n, 299, np.zeros(1000, np.uint8), off.copy(), ln.copy())
    assert np.array_equal(loc, wloc) and used <= z.size and (off % 16 == 0).all()
    got, status = eng.png_decode_z(torch.from_numpy(z[:used]).cuda(), torch.from_numpy(off.view(np.int32)).cuda(),
                                   torch.from_numpy(ln.view(np.int32)).cuda())
    assert not status.cpu().numpy().any()
    assert np.array_equal(got.cpu().numpy(), want)


def _slides_on_disk(tmp_path, n_slides, per, seed):
    from biscuit_amd.synthetic import make_slides
    tiles, sidx, y = make_slides(n_slides, per, seed=seed)
    paths = []
    for i in range(n_slides):
        p = str(tmp_path / f'z{i}.tfrecords')
        tfr.write_slide(p, f'z{i}', tiles[sidx == i])
        paths.append(p)
    return paths, {f'z{i}': int(y[i]) for i in range(n_slides)}


@pytest.mark.parametrize('pooled', [False, True], ids=['one_engine', 'pool_with_decode_cus'])
def test_evaluate_from_compressed_chunks_equals_host_decoded_tiles(tmp_path, pooled, monkeypatch):
    """``slides_from_tfrecords(gpu_decode=True)``: the host copies zlib streams, the device inflates and un-filters -- the same tile
    table and slide results, bit for bit, as the host decoder's tiles; with an ``EnginePool(reserve_cus=16)`` the inflate runs on
    CU-masked streams of its own.  A slide holding a JPEG record goes to the host decoder as a whole.  Small compressed chunks so
    that several are in flight and a slide spans chunks."""
    import io
    from PIL import Image
    from biscuit_amd import inference as inf
    from biscuit_amd.engine import Engine, EnginePool
    monkeypatch.setattr(inf, 'CHUNK_TILES_Z', 8)
    monkeypatch.setattr(inf, 'RAMP_CHUNKS_Z', (3, 5))
    paths, labels = _slides_on_disk(tmp_path, 3, 19, seed=31)
    # a fourth slide with one JPEG among its PNG records
    t = make_tiles(4, seed=8)
    b = io.BytesIO(); Image.fromarray(t[2]).save(b, format='JPEG', quality=92)
    pj = str(tmp_path / 'zj.tfrecords')
    tfr.write_slide(pj, 'zj', [tfr.encode_image(t[0]), tfr.encode_image(t[1]), b.getvalue(), tfr.encode_image(t[3])])
    paths.append(pj); labels['zj'] = 1
    w = synthetic_weights(1)
    if pooled:
        e = EnginePool(w, n_streams=2, reserve_cus=16, dtype='f16', max_batch=16, max_mc=5)
        assert len(e.decode_streams) == 2
    else:
        e = Engine(w, dtype='f16', max_batch=16, max_mc=5)
    a = inf.evaluate(e, inf.slides_from_tfrecords(paths, labels, gpu_decode=True), outcome='cohort', mc_n=5, seed=3, batch=16)
    ref = inf.evaluate(e, inf.slides_from_tfrecords(paths, labels), outcome='cohort', mc_n=5, seed=3, batch=16)
    assert list(a.slide_count) == [19, 19, 19, 4]
    for col in ('cohort-y_pred1', 'cohort-uncertainty1'):
        assert np.array_equal(a.tile_df[col].to_numpy(), ref.tile_df[col].to_numpy()), col
    assert np.array_equal(a.slide_pred, ref.slide_pred) and np.array_equal(a.slide_unc, ref.slide_unc)
    srcs = [s.source for s in inf.slides_from_tfrecords(paths, labels, gpu_decode=True)]
    assert [s.z_ok() for s in srcs] == [True, True, True, False]
    if pooled:
        e.close()


def test_a_damaged_stream_fails_the_run_loudly(tmp_path):
    """One flipped byte inside a tile's zlib stream: the device path raises (from the status words, looked at one chunk late and at
    the end of the run) -- never a silently wrong tile."""
    from biscuit_amd import inference as inf
    from biscuit_amd.engine import Engine
    raws = [bytearray(tfr.encode_image(t)) for t in make_tiles(6, seed=5)]
    at = raws[4].find(b'IDAT') + 2000
    raws[4][at] ^= 0x5a
    path = str(tmp_path / 'd.tfrecords')
    tfr.write_slide(path, 'd', [bytes(r) for r in raws])
    e = Engine(synthetic_weights(1), dtype='f16', max_batch=8, max_mc=2)
    with pytest.raises(IOError, match='device inflate refused 1 tile'):
        inf.evaluate(e, inf.slides_from_tfrecords([path], {'d': 1}, gpu_decode=True), outcome='cohort', mc_n=2, seed=3, batch=8)


def test_mutated_streams_agree_with_zlib_and_terminate(eng):
    """A seeded fuzz: valid streams of every block kind with 1-3 random byte edits (plus truncations) -- the bytes of a damaged file.
    Every lane must come back (the launch ends), and the verdict must be zlib's: status 0 with zlib's bytes where ``decompress``
    returns exactly the expected length, non-zero where it raises or returns another length."""
    px = 24
    n_out = px * (1 + 3 * px)
    rng = np.random.default_rng(2024)
    base = []
    for nm, raw in payloads(px, rng).items():
        for level in (1, 6, 9):
            base.append(zlib.compress(raw, level))
        if nm == 'text':                                          # a 512-byte window
            c = zlib.compressobj(6, zlib.DEFLATED, 9)
            base.append(c.compress(raw) + c.flush())
        else:
            base.append(zlib.compress(raw, 0))                    # stored blocks
    streams = []
    for i in range(1536):
        s = bytearray(base[i % len(base)])
        for _ in range(int(rng.integers(1, 4))):
            s[int(rng.integers(0, len(s)))] = int(rng.integers(0, 256))
        if i % 7 == 0:
            s = s[:int(rng.integers(1, len(s)))]
        if len(s) < 2:
            s = bytearray(b'\x78\x9c')
        streams.append(bytes(s))
    got, status = run(eng, streams, px)
    torch.cuda.synchronize()
    n_ok = 0
    for i, s in enumerate(streams):
        try:
            d = zlib.decompressobj()
            ref = d.decompress(s) + d.flush()
            ok = d.eof and len(ref) == n_out and not d.unused_data and max(ref[::1 + 3 * px]) <= 4
        except zlib.error:
            ok = False
        assert (status[i] == 0) == ok, (i, int(status[i]), ok)
        if ok:
            n_ok += 1
            assert got[i].tobytes() == ref, i
    assert 0 < n_ok < len(streams)          # the corpus holds both survivors and casualties
