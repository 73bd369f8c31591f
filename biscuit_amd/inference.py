"""The per-slide MC-dropout inference loop -- what ``Project.evaluate(model, outcome,
filters, save_predictions=True)`` (``biscuit/experiment.py:917-922``) and the validation
step of ``Project.train(..., save_predictions=True)`` (``experiment.py:1042-1051``) do for
BISCUIT: stream every slide's 299x299 tiles, run the classifier with dropout active for
``uq_n`` passes, keep per-tile mean/std, and reduce to slide-level prediction/uncertainty.

Slides are sharded over ranks (one process per GPU); tiles stream in batches of
``batch`` that may span slide boundaries (a per-tile slide index drives the device-side
segmented reduce); each tile's Philox counter is its GLOBAL index in dataset order, so
results do not depend on batch size, sharding or rank count.
"""
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import pandas as pd
import torch

from . import distributed as D
from .predictions import (EVAL_NAME, TableWriter, assemble_shards, remove_stale_shards, save_tile_predictions, shard_name,
                          tile_frame, write_shard_index)


@dataclass
class Slide:
    """One slide = one TFRecord's worth of tiles.  ``tiles`` is either a uint8 array
    [T,299,299,3] (host or device) or a zero-argument callable returning one.  ``source`` (optional): an object with
    ``read(first, count, out)`` that decodes tiles [first, first + count) into a caller-supplied uint8 buffer and a
    ``rows`` flag (``TFRecordSource``): ``evaluate`` then streams the slide in fixed chunks through a ring of reusable
    pinned buffers instead of calling ``tiles``."""
    name: str
    tiles: object
    n_tiles: int
    y_true: int = 0
    patient: Optional[str] = None
    loc: Optional[np.ndarray] = None
    source: Optional[object] = None

    def load(self):
        t = self.tiles() if callable(self.tiles) else self.tiles
        return t


class PngRows:
    """A slide's tiles as the tile reader leaves them when the GPU reverses the PNG scanline filters: uint8
    [T,px,1+3*px] (host, usually pinned).  ``evaluate`` copies them to the device and calls ``Engine.png_unfilter``."""
    def __init__(self, rows):
        self.rows = rows


class TFRecordSource:
    """Chunk-wise decoder of one slide's TFRecord for ``evaluate``'s pinned ring: ``read(first, count, out)`` decodes
    tiles [first, first + count) into ``out`` (a uint8 numpy view, usually of page-locked memory) on the reader's thread
    pool.  ``rows``: stop at the inflated PNG scanlines ([count,px,1+3*px]; the GPU reverses the filters).  A slide with
    a record outside the native decoders' subset is decoded whole with Pillow once and served from memory."""
    def __init__(self, path, n_tiles, tile_px=299, rows=False, z=False):
        self.path, self.n_tiles, self.tile_px, self.rows = path, int(n_tiles), int(tile_px), bool(rows)
        # z: hand the tiles over COMPRESSED (``read_z``: the records' zlib streams, packed; the device inflates them -- bq_png_inflate);
        # decided per slide before its first chunk: a slide with a record that is not an 8-bit RGB PNG tile stays on ``read``
        self.z = bool(z)
        self._reader = None
        self._fallback = None
        self._probed = False                    # the once-per-slide decoder decision (``read``) has been taken

    def z_ok(self):
        """True when the whole slide can go the compressed way (every record an 8-bit RGB, non-interlaced PNG of the tile size)."""
        from . import tfrecord_native as tn
        # (larger tiles: the un-filter kernel takes rows of up to 1 024 bytes = 341 px; those slides stay on the host decoder)
        if not (self.z and tn.available() and self.n_tiles and self.tile_px <= 341):
            return False
        if self._reader is None:
            self._reader = tn.NativeReader(self.path)
        probe = np.zeros(1, np.uint8)
        off, ln = np.zeros(CHUNK_TILES_Z, np.uint32), np.zeros(CHUNK_TILES_Z, np.uint32)
        for first in range(0, self.n_tiles, CHUNK_TILES_Z):      # (chunk-wise: the offsets of one call are 32-bit)
            try:
                self._reader.extract_z(first, min(CHUNK_TILES_Z, self.n_tiles - first), self.tile_px, probe, off, ln)
            except MemoryError:
                continue                                 # (headers fine, only the buffer was too small: as intended)
            except (tn.UnsupportedImage, ValueError, IOError):
                return False
        return True

    def read_z(self, first, count, z, off, length):
        """The zlib streams of tiles [first, first + count) packed into ``z`` (uint8), offsets / lengths into ``off`` / ``length``
        (uint32 [count]).  Returns the bytes used; MemoryError (bytes needed in ``.args[1]``) when ``z`` is too small."""
        return self._reader.extract_z(first, count, self.tile_px, z, off, length)[0]

    def chunk_shape(self, count):
        px = self.tile_px
        return (count, px, 1 + 3 * px) if self.rows else (count, px, px, 3)

    def read(self, first, count, out):
        from . import tfrecord, tfrecord_native as tn
        if self._fallback is None and tn.available():
            if self._reader is None:
                self._reader = tn.NativeReader(self.path)
            if not self._probed:
                # ONE decoder per slide, decided before its first chunk: a record the native decoders refuse (a progressive
                # JPEG, ...) sends the WHOLE slide to Pillow -- as the whole-slide loader does (Slide.load) -- instead of the chunks
                # from that record on: the native islow IDCT and Pillow's libjpeg-turbo are not bound to agree to the last bit.
                # (Its own flag: ``z_ok`` may have opened the reader already -- round 5 skipped the probe then.)
                self._probed = True
                if self._reader.probe(self.tile_px) is not None:
                    self._reader.close()
                    self._reader = None
                    self._fallback = tfrecord.read_slide(self.path, self.tile_px, rows=self.rows)[1]
            if self._reader is not None:
                try:
                    self._reader.decode(first, count, self.tile_px, out=out, rows=self.rows)
                    return
                except tn.UnsupportedImage:
                    # behind a clean probe only a damaged entropy-coded stream ends here; chunks of this slide went out already
                    if first > 0:
                        raise
        if self._fallback is None:                      # Pillow (or the pure-Python reader), the whole slide once
            self._fallback = tfrecord.read_slide(self.path, self.tile_px, rows=self.rows)[1]
        out[...] = self._fallback[first:first + count]

    def close(self):
        if self._reader is not None:
            self._reader.close()
            self._reader = None
        self._fallback = None
        self._probed = False


def pick_unfilter_mode(path, tile_px=299, sample=48):
    """'auto' for ``slides_from_tfrecords``: decode the first ``sample`` tiles of one slide both ways and keep the GPU
    un-filter only where it pays -- the host alone is slower than the GPU consumes tiles (~28 k/s) AND stopping at the
    scanlines makes it at least 8 % faster (noise-like synthetic tiles: Sub / Up rows, cheap on the host; photo-like
    tiles: +20-30 %).  Returns (use_gpu_unfilter, host tiles/s, rows tiles/s)."""
    import time
    from . import tfrecord_native as tn
    if not tn.available():
        return False, 0.0, 0.0
    try:
        with tn.NativeReader(path) as r:
            n = min(sample, len(r))
            if n == 0:
                return False, 0.0, 0.0
            r.decode(0, n, tile_px)                      # page cache, thread pool warm
            t0 = time.perf_counter(); r.decode(0, n, tile_px); t_full = time.perf_counter() - t0
            t0 = time.perf_counter(); r.decode(0, n, tile_px, rows=True); t_rows = time.perf_counter() - t0
    except (tn.UnsupportedImage, ValueError, IOError):
        return False, 0.0, 0.0
    full, rows = n / max(t_full, 1e-9), n / max(t_rows, 1e-9)
    return (full < 26000.0 and rows > 1.08 * full), full, rows


def slides_from_tfrecords(paths, labels, patients=None, tile_px=299, pinned=None, gpu_unfilter=None, gpu_decode=False):
    """One ``Slide`` per ``*.tfrecords`` file (Slideflow writes one file per slide).  Tiles are
    decoded lazily when the slide's turn comes (``evaluate`` decodes one slide ahead on a host thread);
    only the record headers are scanned up front.  labels: {slide name (file stem): 0/1}.
    ``pinned`` (default: when a GPU is present) decodes into page-locked memory so the H2D copy is
    asynchronous and overlaps the next slide's decode.  ``gpu_unfilter`` (default off; tiles up to 341 px): the host stops at
    the inflated PNG scanlines and the GPU reverses their filters (``Engine.png_unfilter``) -- 12-30 % more tiles per host
    core for 0.4-0.6 ms of GPU time per launch of up to 512 tiles (host side): for hosts whose cores,
    not the GPU, bound the run."""
    import os
    from . import tfrecord
    if pinned is None:
        pinned = torch.cuda.is_available()
    if gpu_unfilter == 'auto':                          # a short measurement on the first slide decides
        gpu_unfilter = bool(paths) and tile_px <= 341 and torch.cuda.is_available() and pick_unfilter_mode(paths[0], tile_px)[0]
    gpu_unfilter = bool(gpu_unfilter)
    # gpu_decode (round 5): the host only walks the record framing and copies the PNG tiles' zlib streams; the GPU inflates them
    # (one stream per lane, on compute units an ``EnginePool(reserve_cus=...)`` keeps out of the inference streams' masks) and
    # reverses the scanline filters.  For hosts with few cores per GPU: 16 CUs inflate 27-39 k tiles/s (profiles/r05_inflate.txt).
    gpu_decode = bool(gpu_decode) and bool(pinned)
    out = []
    for path in paths:
        name = os.path.splitext(os.path.basename(path))[0]
        count = tfrecord.count_records(path)

        def loader(pth=path, n=count):
            if gpu_unfilter and n:
                t = torch.empty((n, tile_px, 1 + 3 * tile_px), dtype=torch.uint8, pin_memory=bool(pinned))
                tfrecord.read_slide(pth, tile_px, out=t.numpy(), rows=True)
                return PngRows(t)
            if pinned and n:
                t = torch.empty((n, tile_px, tile_px, 3), dtype=torch.uint8, pin_memory=True)
                tfrecord.read_slide(pth, tile_px, out=t.numpy())
                return t
            return tfrecord.read_slide(pth, tile_px)[1]
        out.append(Slide(name, loader, count, y_true=int(labels.get(name, 0)),
                         patient=(patients or {}).get(name),
                         source=TFRecordSource(path, count, tile_px, rows=gpu_unfilter and not gpu_decode, z=gpu_decode) if pinned else None))
    return out


@dataclass
class EvalResult:
    tile_df: Optional[pd.DataFrame]          # this rank's tile rows (Slideflow headers)
    slide_names: List[str]
    slide_pred: np.ndarray                   # float64 [S], all slides (after the gather)
    slide_unc: np.ndarray
    slide_count: np.ndarray
    slide_y_true: np.ndarray
    local_slides: List[int] = field(default_factory=list)
    table_path: Optional[str] = None         # the tile table on disk: THE table (world 1; rank 0 after the splice) or this rank's shard
    table_rows: int = 0                      # rows this rank wrote
    f16_headroom: float = float('inf')       # minimum over the run's range checks of 65504 / max |stored activation| (f16 engines)
    f16_checks: int = 0

    def slide_frame(self, pred_thresh=0.5, level='slide'):
        """Group table in ``process_group_predictions`` form from the device-reduced means."""
        from .threshold import group_frame
        keep = self.slide_count > 0
        names = [n for n, k in zip(self.slide_names, keep) if k]
        return group_frame(names, self.slide_pred[keep], self.slide_y_true[keep].astype(np.uint8),
                           self.slide_unc[keep], pred_thresh, level)


CHUNK_TILES = 512        # tiles per pinned buffer: 137 MB at 299 px (a 10^4-tile slide is twenty chunks, never one allocation)
RAMP_CHUNKS = (128, 256) # the first chunks of a run are short: the GPU starts after 128 decoded tiles (4 ms of the decoder), not 512
CHUNK_TILES_Z = 4096     # compressed chunks (gpu_decode): one zlib stream per LANE, so a chunk is what keeps the decode CUs' waves full
RAMP_CHUNKS_Z = (512, 1024, 2048)
Z_SLOT_MAX = 1 << 30      # bytes of one pinned slot of the compressed ring at most (three slots are leased)
Z_FRACTION = 0.9         # pinned bytes per tile of a compressed chunk, as a fraction of the raw scanlines (a nearly incompressible 299-px
                         # PNG: 227 KB of 268 KB = 0.85; a photograph-like one 0.58); a chunk that does not fit is cut in two
RING_SLOTS = 3
PREFETCH_CHUNKS = 2      # decoded chunks waiting for the GPU (plus the one being decoded)


class _PinnedRing:
    """RING_SLOTS reusable page-locked buffers (page-locking 137 MB costs tens of milliseconds, which round 3 paid per slide).  A
    slot is free again when the event recorded behind its last host-to-device copy has completed.  Rings are LEASED: one
    ``evaluate`` call holds a ring from its first chunk to its last and hands it back, so calls that follow each other reuse the
    same pinned memory and calls that overlap (two threads of one process) never share slots.  A ring is sized for the larger
    of the two chunk layouts of its tile size -- decoded tiles (px * px * 3 bytes) and filtered PNG scanlines (px * (1 + 3 px))
    -- so alternating the two modes does not re-allocate it (round 4 did: one size in the cache at a time)."""
    _idle = []                       # rings not leased at the moment
    _lock = None
    MAX_IDLE = 2

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.bufs = [torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(RING_SLOTS)]
        self.events = [None] * RING_SLOTS
        self.next = 0

    @staticmethod
    def chunk_bytes(tile_px):
        return CHUNK_TILES * tile_px * (1 + 3 * tile_px)

    @classmethod
    def lease(cls, nbytes):
        import threading
        if cls._lock is None:
            cls._lock = threading.Lock()
        with cls._lock:
            for i, r in enumerate(cls._idle):
                if r.nbytes >= nbytes:
                    return cls._idle.pop(i)
            cls._idle.clear()                        # (a larger tile size: the smaller buffers go)
        return cls(nbytes)

    def release(self):
        for i, ev in enumerate(self.events):
            if ev is not None:
                ev.synchronize()
                self.events[i] = None
        with self._lock:
            if len(self._idle) < self.MAX_IDLE:
                self._idle.append(self)

    def acquire(self):
        i = self.next
        self.next = (i + 1) % RING_SLOTS
        if self.events[i] is not None:
            self.events[i].synchronize()             # (the feeder thread waits, not the thread that launches kernels)
            self.events[i] = None
        return i


def _feed_chunks(slides, mine, dev, copy_stream):
    """Generator over this rank's slides in order: yields (li, si, first, count, tensor, event, is_rows).  Slides with a
    ``source`` are decoded chunk by chunk on a feeder thread into the pinned ring and copied to the device on
    ``copy_stream`` (event = the copy's completion; the tensor was allocated on that stream); PREFETCH_CHUNKS chunks may
    wait decoded and copied while the GPU works -- with 512-tile chunks that is two 1 000-tile slides of lead.  Other
    slides are loaded whole on the same thread and handed over as they are (event None)."""
    import queue
    import threading
    q = queue.Queue(maxsize=PREFETCH_CHUNKS)
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    zc = None                                            # the compressed chunk being filled (gpu_decode), if any

    def emit_z():
        nonlocal zc
        c, zc = zc, None
        if c is None or not c['segs']:
            return True
        total = c['hdr'] + c['pos']
        with torch.cuda.stream(copy_stream):
            d = torch.empty(total, dtype=torch.uint8, device=dev)
            d.copy_(c['ring'].bufs[c['slot']][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        c['ring'].events[c['slot']] = ev
        return put((c['segs'], c['cap'], 0, c['n'], d, ev, 'z'))

    def work():
        nonlocal zc
        ring = None
        n_chunks = 0                                     # chunks decoded so far in this run (over all slides)
        try:
            for li, si in enumerate(mine):
                s = slides[si]
                src = getattr(s, 'source', None)
                if src is None or s.n_tiles == 0 or copy_stream is None:
                    if not emit_z():
                        return
                    if copy_stream is None:
                        item = (li, si, 0, s.n_tiles, s.load(), None, False)
                    else:
                        # a loader may launch GPU work of its own (tiles resident on the device): on this thread that must
                        # not be the legacy default stream -- a barrier across the pool's streams, and unordered against the
                        # consumer's -- but the copy stream, with an event for the consumer to wait on
                        with torch.cuda.stream(copy_stream):
                            loaded = s.load()
                            ev = None
                            if torch.is_tensor(loaded) and loaded.is_cuda:
                                ev = torch.cuda.Event()
                                ev.record(copy_stream)
                        item = (li, si, 0, s.n_tiles, loaded, ev, False)
                    if not put(item):
                        return
                    continue
                if getattr(src, 'z', False) and src.z_ok():
                    # compressed chunks: [off u32[cap] | len u32[cap] | packed zlib streams] in one pinned slot, one H2D copy.  A
                    # chunk runs ACROSS slides (the device inflates one stream per lane: a 1 000-tile slide alone would leave the
                    # decode CUs' waves mostly empty); item = (segments [(li, si, first, count)], cap, 0, tiles, buffer, event, 'z')
                    px = src.tile_px
                    need = min(CHUNK_TILES_Z * (8 + int(Z_FRACTION * px * (1 + 3 * px))) + 64, Z_SLOT_MAX)   # (larger tiles: chunks of fewer)
                    if ring is None or ring.nbytes < need:
                        if not emit_z():
                            return
                        if ring is not None:
                            ring.release()
                        ring = _PinnedRing.lease(need)
                    try:
                        first = 0
                        while first < s.n_tiles:
                            if zc is None:
                                cap = RAMP_CHUNKS_Z[n_chunks] if n_chunks < len(RAMP_CHUNKS_Z) else CHUNK_TILES_Z
                                n_chunks += 1
                                slot = ring.acquire()
                                zc = {'slot': slot, 'buf': ring.bufs[slot].numpy(), 'cap': cap, 'hdr': (8 * cap + 15) & ~15, 'n': 0,
                                      'pos': 0, 'segs': [], 'px': px, 'ring': ring}
                            elif zc['px'] != px:
                                if not emit_z():
                                    return
                                continue
                            k, cap, buf = zc['n'], zc['cap'], zc['buf']
                            cnt = min(cap - k, s.n_tiles - first)
                            off = buf[:4 * cap].view(np.uint32)[k:]
                            ln = buf[4 * cap:8 * cap].view(np.uint32)[k:]
                            room = buf[zc['hdr'] + zc['pos']:ring.nbytes]
                            while cnt:
                                try:
                                    used = src.read_z(first, cnt, room, off, ln)
                                    break
                                except MemoryError:
                                    if cnt == 1 and not zc['segs']:
                                        raise
                                    cnt //= 2                      # (tiles that compress worse than the slot was sized for)
                            if cnt:
                                off[:cnt] += np.uint32(zc['pos'])
                                zc['segs'].append((li, si, first, cnt))
                                zc['n'] += cnt
                                zc['pos'] += used
                                first += cnt
                            if not cnt or zc['n'] == cap:
                                if not emit_z():
                                    return
                    finally:
                        src.close()
                    continue
                if not emit_z():                                   # (a slide that goes the decoded way: what is open goes first)
                    return
                per = int(np.prod(src.chunk_shape(1)))
                px = getattr(src, 'tile_px', None)
                need = max(CHUNK_TILES * per, _PinnedRing.chunk_bytes(px) if px else 0)
                if ring is None or ring.nbytes < need:
                    if ring is not None:
                        ring.release()
                    ring = _PinnedRing.lease(need)
                try:
                    first = 0
                    while first < s.n_tiles:
                        size = RAMP_CHUNKS[n_chunks] if n_chunks < len(RAMP_CHUNKS) else CHUNK_TILES
                        n_chunks += 1
                        cnt = min(size, s.n_tiles - first)
                        slot = ring.acquire()
                        host = ring.bufs[slot][:cnt * per].view(src.chunk_shape(cnt))
                        src.read(first, cnt, host.numpy())
                        with torch.cuda.stream(copy_stream):
                            d = torch.empty(src.chunk_shape(cnt), dtype=torch.uint8, device=dev)
                            d.copy_(host, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        ring.events[slot] = ev
                        if not put((li, si, first, cnt, d, ev, src.rows)):
                            return
                        first += cnt
                finally:
                    src.close()
            if emit_z():
                put(None)
        except BaseException as e:                   # noqa: BLE001 -- re-raised in the consumer
            put(e)
        finally:
            if ring is not None:
                ring.release()

    th = threading.Thread(target=work, name='bq-tile-feeder', daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        th.join(timeout=30)


def _take_front(parts, k, cat):
    """(the first k rows of the arrays in ``parts``, what is left of the list): views where possible, ``cat`` only over the pieces of a
    front that spans several arrays."""
    out, i = [], 0
    while k > 0:
        t = parts[i]
        if t.shape[0] <= k:
            out.append(t); k -= t.shape[0]; i += 1
        else:
            out.append(t[:k]); parts = parts[:i] + [t[k:]] + parts[i + 1:]; k = 0
    return (out[0] if len(out) == 1 else cat(out)), parts[i:]


def _to_device(t, device):
    if torch.is_tensor(t):
        return t.to(device=device, dtype=torch.uint8, non_blocking=True).contiguous()
    return torch.from_numpy(np.ascontiguousarray(t)).to(device, non_blocking=True)


class _TableStream:
    """The tile table written WHILE the GPU works (round 6; Slideflow -- and rounds 1-5 here -- wrote it with one
    ``DataFrame.to_csv`` after the last batch: 1.75 s per 200 000 rows, serial, behind 6 s of GPU time).  ``submit`` takes a
    batch's results as they were enqueued -- one device tensor [2, n, 2] (mean | std), the event behind the batch's last kernel,
    and the runs of tiles per slide it holds --; a host thread waits for the event on a side stream, copies the 4 KB to pinned
    memory there and appends the rows through ``TableWriter`` (libbiscuit_io; the GIL is released while it formats and writes).
    Batches are written in submission order, so the rows are in dataset order however many batches are in flight.  For a shard
    of a multi-rank run it also notes every slide's byte range (``write_shard_index``)."""

    def __init__(self, path, outcome, with_loc, dev, max_batch, shard=None):
        import queue
        import threading
        self.writer = TableWriter(path, outcome, with_loc)
        self.path, self.outcome, self.with_loc, self.shard = path, outcome, with_loc, shard
        self.dev = torch.device(dev)
        self.cuda = self.dev.type == 'cuda'
        self.side = torch.cuda.Stream(device=self.dev) if self.cuda else None
        self.host = torch.empty((2 * max_batch * 2,), dtype=torch.float32, pin_memory=self.cuda)
        self.q = queue.Queue(maxsize=64)
        self.error = None
        self.index = []                      # [global slide index, name, rows, offset, length]
        self.rows = 0
        self.th = threading.Thread(target=self._work, name='bq-table-writer', daemon=True)
        self.th.start()

    def _work(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    return
                if self.error is not None:
                    continue                 # (drain: the producer must never block on a dead writer)
                out2, ev, segs = item
                n = out2.shape[1]
                if self.cuda:
                    host = self.host[:4 * n].view(2, n, 2)
                    with torch.cuda.stream(self.side):
                        if ev is not None:
                            self.side.wait_event(ev)
                        host.copy_(out2, non_blocking=True)
                    self.side.synchronize()
                    arr = host.numpy()
                else:
                    arr = out2.numpy()
                at = 0
                for si, name, y_true, loc, count in segs:
                    t0 = self.writer.tell()
                    self.writer.rows(name, y_true, arr[0, at:at + count], arr[1, at:at + count], loc)
                    t1 = self.writer.tell()
                    if self.index and self.index[-1][0] == si:
                        self.index[-1][2] += count
                        self.index[-1][4] += t1 - t0
                    else:
                        self.index.append([si, name, count, t0, t1 - t0])
                    at += count
                    self.rows += count
                del out2, item
        except BaseException as e:           # noqa: BLE001 -- re-raised by the producer
            self.error = e
            while self.q.get() is not None:  # keep draining until the producer says stop
                pass

    def submit(self, out2, ev, segs):
        if self.error is not None:
            self.finish()
        if self.cuda:
            out2.record_stream(self.side)
        self.q.put((out2, ev, segs))

    def abort(self):
        """The run failed elsewhere: stop the thread and close the file (what is on disk stays, without an index)."""
        if self.th is not None:
            self.q.put(None)
            self.th.join()
            self.th = None
        self.error = None
        try:
            self.writer.close()
        except IOError:
            pass

    def finish(self):
        """Everything submitted is on disk and the file is closed when this returns; raises what the writer thread met."""
        if self.th is not None:
            self.q.put(None)
            self.th.join()
            self.th = None
        err, self.error = self.error, None
        if err is not None:
            try:
                self.writer.close()
            finally:
                raise err
        rows, _ = self.writer.close()
        if self.shard is not None:
            write_shard_index(self.path, self.shard[0], self.shard[1], self.outcome, self.with_loc, self.index)
        return rows


def evaluate(engine, slides: Sequence[Slide], outcome='cohort', mc_n=None, seed=None, batch=256,
             mc_mode='head', tile_uq=None, save_dir=None, keep_tiles=True, rank=0, world=1, norm_fit=None,
             table_name=EVAL_NAME, table_writer='native', headroom_every=200, headroom_min=2.0):
    """Run MC-dropout inference over ``slides`` and return tile- and slide-level results.

    Every rank passes the SAME slide list; rank r processes ``partition_slides(...)[r]``.
    The slide-level arrays are all-gathered (one collective).

    ``save_dir``: the tile table -- the product ``biscuit.threshold`` reads (experiment.py:688-699) -- is written there as
    ``table_name`` WHILE the GPU works (``_TableStream``; ``keep_tiles`` is not needed for it).  With ``world`` > 1 every rank
    streams its shard ``tile_predictions_eval.rankR.csv`` (+ a byte index of its slides) and closes it BEFORE the all-gather, so
    the gather doubles as "all shards complete"; rank 0 then splices them into the ONE table in dataset order -- byte for byte
    the file a single-rank run writes.  ``table_writer='pandas'`` (or a ``.parquet.gzip`` name) writes with pandas after the run
    instead: the checker of the native writer, and the parquet form.

    ``headroom_every`` (f16 engines only; 0 / None: off): every that many batches -- and on the first -- the eight range taps of
    ``Engine.f16_headroom`` run on up to eight tiles of the batch, behind it on its stream, with no host synchronisation: the
    maxima are copied out asynchronously and looked at when the next check is due (and at the end).  A value at the clamp, or
    less than ``headroom_min`` x of range left, raises ``F16RangeError``: the calibration batch at the start of a run says
    nothing about the 10^5 tiles behind it, and f16's clamp is silent.  ``EvalResult.f16_headroom`` = the minimum seen.

    ``norm_fit`` (``{'target_means': [3], 'target_stds': [3]}``, the block of that name in the model's
    params.json) switches on the `reinhard_fast` stain normaliser of hp.py:19 in front of the staging
    kernel, where results.py:251-252 applies it."""
    hp = engine.hp
    mc_n = int(mc_n or hp.uq_n)
    seed = int(hp.seed if seed is None else seed)
    counts = [s.n_tiles for s in slides]
    parts = D.partition_slides(counts, world)
    mine = parts[rank]
    offsets = D.global_tile_offsets(counts)
    dev = engine.device
    n_local = len(mine)
    # an EnginePool alternates batches over independent contexts / HIP streams
    pool = engine if hasattr(engine, 'engines') else None
    engines = pool.engines[:len(pool)] if pool else [engine]      # len(pool) = batches in flight
    acc = [None] * len(engines)
    n_batches = 0
    rows_mean, rows_std, rows_slide, rows_true, rows_loc = [], [], [], [], []
    with_loc = any(s.n_tiles for s in slides) and all(s.loc is not None for s in slides if s.n_tiles)    # (every rank decides the same: one header)
    native = save_dir is not None and table_writer == 'native' and table_name.endswith('.csv')
    if table_writer not in ('native', 'pandas'):
        raise ValueError(f"table_writer must be 'native' or 'pandas', not {table_writer!r}")
    if save_dir is not None and not native and not keep_tiles:
        raise ValueError('the pandas writer needs keep_tiles=True (it writes the frame after the run)')
    table = None
    if save_dir is not None and rank == 0:
        import os
        if os.path.isdir(save_dir):
            remove_stale_shards(save_dir, world if world > 1 else 0, table_name)      # (a one-rank run leaves THE table only)
    if native:
        import os
        tpath = os.path.join(save_dir, table_name if world == 1 else shard_name(table_name, rank))
        table = _TableStream(tpath, outcome, with_loc, dev, batch, shard=None if world == 1 else (rank, world))
    pend_segs = []                           # (slide index, name, y_true, loc rows or None, count) of the pending tiles, in order
    # the f16 range monitor (see the docstring): checks in flight = (batch number, first global tile index, pinned [8, 2], event)
    monitor = bool(headroom_every) and all(getattr(e, 'dtype', None) == 'f16' and hasattr(e, 'f16_headroom_async') for e in engines)
    hr_pending, hr_state = [], {'min': float('inf'), 'checks': 0}

    def headroom_look(block):
        from .engine import Engine, F16RangeError
        while hr_pending and (block or hr_pending[0][3].query()):
            nb, g0, host, ev = hr_pending.pop(0)
            ev.synchronize()
            a = host.numpy()
            hr_state['checks'] += 1
            worst = int(np.argmax(a[:, 0]))
            hr = 65504.0 / max(float(a[worst, 0]), 1e-30)
            hr_state['min'] = min(hr_state['min'], hr)
            if a[:, 1].sum() > 0 or hr < float(headroom_min):
                sat = {Engine.HEADROOM_TAPS[i][0]: int(a[i, 1]) for i in range(a.shape[0]) if a[i, 1] > 0}
                raise F16RangeError(
                    f'f16 storage at its range limit in batch {nb} (global tile {g0} on): ' +
                    (f'{sat} values clamped at +-65504' if sat else f'only {hr:.2f}x of range left at {Engine.HEADROOM_TAPS[worst][0]} '
                     f'(peak {float(a[worst, 0]):.4g}; headroom_min {headroom_min})') +
                    '; the results from there on would be plausible and wrong.  Re-run with Engine.calibrate() on tiles like these, '
                    'or with dtype bf16 / f32')

    # stream tiles of this rank's slides in batches that may span slides
    pend_tiles, pend_sidx, pend_gidx = [], [], []
    pend_n = 0

    def flush(final=False):
        nonlocal pend_tiles, pend_sidx, pend_gidx, pend_n, n_batches
        while pend_n >= batch or (final and pend_n > 0):
            take = min(batch, pend_n)
            # the first `take` rows of what is pending: a VIEW when they lie in one tensor, one batch-sized copy when the batch spans
            # two (round 4 concatenated everything pending -- a 1 000-tile slide behind a 200-tile remainder: 330 MB copied to cut 256
            # tiles off the front, 0.11 ms per batch in config 3's trace)
            cur, pend_tiles = _take_front(pend_tiles, take, torch.cat)
            cs, pend_sidx = _take_front(pend_sidx, take, torch.cat)
            cg, pend_gidx = _take_front(pend_gidx, take, np.concatenate)
            cur, cs = cur.contiguous(), cs.contiguous()
            # global tile indices inside a batch are contiguous per slide but not across
            # slides: run one bq_mc_infer per contiguous run so the Philox counter is exact
            out2 = torch.empty((2, take, 2), dtype=torch.float32, device=dev)       # mean | std: ONE device-to-host copy per batch
            mean, std = out2[0], out2[1]
            brk = np.flatnonzero(np.diff(cg) != 1) + 1
            starts = np.concatenate([[0], brk]); ends = np.concatenate([brk, [take]])
            k = n_batches % len(engines)

            # a batch that spans slides holds tiles whose global indices are not one consecutive run: the backbone does not care,
            # the head's Philox counter does -- it takes the indices as an array then (bq_set_tile_index_array): ONE launch sequence
            # per batch whatever its composition (round 4: one head call per run, 4 x the head time with 64-tile slides), and a
            # tile's result does not depend on batch size, sharding or rank count
            gdev = None
            if len(starts) > 1:
                gdev = torch.from_numpy(np.ascontiguousarray(cg)).to(dev, non_blocking=True) if torch.device(dev).type == 'cuda' \
                    else torch.from_numpy(np.ascontiguousarray(cg))

            if monitor and hr_pending and n_batches % int(headroom_every) == 0:
                headroom_look(block=True)            # the previous check: one interval old, long finished -- a run fails one interval late at most

            def work(eng, cur=cur, gdev=gdev):
                if norm_fit is not None:
                    cur = eng.reinhard_fast(cur, norm_fit['target_means'], norm_fit['target_stds'])
                if gdev is None:
                    eng.mc_infer(cur, mc_n, seed, tile_idx0=int(cg[0]), mc_mode=mc_mode, out=(mean, std))
                else:
                    eng.mc_infer(cur, mc_n, seed, tile_idx0=0, mc_mode=mc_mode, out=(mean, std), tile_idx=gdev)
                acc[k] = eng.slide_reduce(mean, std, cs, max(n_local, 1), tile_uq=tile_uq, acc=acc[k])
                if monitor and n_batches % int(headroom_every) == 0:
                    hr = eng.f16_headroom_async(cur)
                    host = torch.empty(hr.shape, dtype=torch.float32, pin_memory=True)
                    host.copy_(hr, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    hr_pending.append((n_batches, int(cg[0]), host, ev))
            if pool:
                # these tensors were allocated on the caller's stream and are read on the pool's: tell the
                # caching allocator, or the next batch's temporaries may reuse their memory while this
                # batch's kernels are still in flight
                st = getattr(pool, 'streams', None)
                if st and cur.is_cuda:
                    for t in (cur, cs, out2) + ((gdev,) if gdev is not None else ()):
                        t.record_stream(st[k])
                pool.run(n_batches, work, wait_for_current=True)
            else:
                work(engine)
            if table is not None:
                ev = None
                if out2.is_cuda:
                    ev = torch.cuda.Event()
                    st = getattr(pool, 'streams', None) if pool else None
                    ev.record(st[k] if st else torch.cuda.current_stream(dev))
                segs, left = [], take
                while left:
                    si, name, yt, loc, c = pend_segs[0]
                    if c <= left:
                        segs.append(pend_segs.pop(0)); left -= c
                    else:
                        segs.append((si, name, yt, None if loc is None else loc[:left], left))
                        pend_segs[0] = (si, name, yt, None if loc is None else loc[left:], c - left)
                        left = 0
                table.submit(out2, ev, segs)
            n_batches += 1
            if keep_tiles:          # device tensors; copied to the host once everything has been enqueued
                rows_mean.append(mean); rows_std.append(std)
            pend_n -= take

    # With a pool, everything this function itself enqueues (H2D copies, concatenations, a device-side loader) goes to
    # a side stream, not to the default stream: the pool's CU-masked streams are ordinary (blocking) HIP streams, and
    # an operation on the legacy default stream is a barrier across all of those -- one such operation per batch and
    # the batches in flight never overlap (measured: 14.9 k tiles/s instead of 24 k through this function).
    prep = torch.cuda.Stream(device=dev) if (pool and torch.device(dev).type == 'cuda') else None
    if prep is not None:
        prep.wait_stream(torch.cuda.current_stream(dev))        # the caller's tensors were made there

    def on_prep(fn):
        def call():
            if prep is None:
                return fn()
            with torch.cuda.stream(prep):
                return fn()
        return call
    # Tiles arrive through a feeder thread: slides with a chunk source (TFRecords) are decoded 512 tiles at a time into a
    # ring of three reusable pinned buffers and copied to the device on a stream of their own, two chunks ahead of the
    # GPU; the native decoder releases the GIL, so decode, H2D copy and kernels overlap.  (Round 3 allocated a fresh
    # page-locked tensor per slide and prefetched one slide.)
    import contextlib
    copy_stream = torch.cuda.Stream(device=dev) if torch.device(dev).type == 'cuda' else None

    # gpu_decode: compressed chunks are inflated on the pool's decode streams (CU-masked: the compute units it keeps out of the
    # inference streams; without a pool, the current stream), round-robin, each with its own table scratch; the status words are
    # looked at one chunk late (check_z)
    z_state = {'k': 0, 'scratch': {}, 'status': []}

    def check_z(block):
        while z_state['status'] and (block or z_state['status'][0][2].query()):
            segs, status, ev2 = z_state['status'].pop(0)
            ev2.synchronize()
            bad = np.flatnonzero(status.numpy()).tolist()
            if bad:
                at, where = 0, None
                for (_, si, first, c) in segs:
                    if at <= bad[0] < at + c:
                        where = f'{slides[si].name}, tile {first + bad[0] - at}'
                    at += c
                raise IOError(f'the device inflate refused {len(bad)} tile(s) (first: {where}, status {int(status[bad[0]])}): damaged PNG '
                              f'data; decode on the host (gpu_decode=False) to see the decoder\'s own error')

    def decode_z(buf, cap, count, ev, px, segs):
        eng0 = engines[0]
        dstreams = getattr(pool, 'decode_streams', None) if pool else None
        main = torch.cuda.current_stream(dev)
        dec = dstreams[z_state['k'] % len(dstreams)] if dstreams else main
        z_state['k'] += 1
        off = buf[:4 * count].view(torch.int32)
        ln = buf[4 * cap:4 * cap + 4 * count].view(torch.int32)
        z = buf[(8 * cap + 15) & ~15:]
        dec.wait_event(ev)
        buf.record_stream(dec)
        key = dec.cuda_stream
        with torch.cuda.stream(dec):
            if key not in z_state['scratch'] or z_state['scratch'][key].numel() < eng0._lib.bq_png_inflate_scratch_bytes(count):
                z_state['scratch'][key] = eng0.inflate_scratch(max(count, CHUNK_TILES_Z))
            rows, status = eng0.png_inflate(z, off, ln, px=px, scratch=z_state['scratch'][key])
            done = torch.cuda.Event()
            done.record(dec)
        main.wait_event(done)
        rows.record_stream(main)
        # the status words leave the device behind the inflate (a few KB, pinned, on the decode stream) and are looked at when the NEXT
        # chunk arrives -- one chunk late, without stalling anything: a damaged stream stops the run there instead of after it
        host = torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
        with torch.cuda.stream(dec):
            host.copy_(status, non_blocking=True)
            ev2 = torch.cuda.Event()
            ev2.record(dec)
        check_z(block=False)
        z_state['status'].append((segs, host, ev2))
        return eng0.png_unfilter_strided(rows, px=px)

    def stream_slides():
        nonlocal pend_n, rows_slide, rows_true
        cur_stream = (lambda: torch.cuda.current_stream(dev)) if copy_stream is not None else None
        def push(li, si, first, count, t):
            nonlocal pend_n, rows_slide, rows_true
            s = slides[si]
            if count == 0:
                return
            if t is not None:                        # (None: the tiles are part of a tensor that is already in the list)
                assert t.shape[0] == count, (s.name, t.shape, count)
                pend_tiles.append(t)
            pend_sidx.append(torch.full((count,), li, dtype=torch.int32, device=dev))
            pend_gidx.append(offsets[si] + first + np.arange(count, dtype=np.int64))
            pend_n += count
            if table is not None:
                pend_segs.append((si, s.name, int(s.y_true), np.asarray(s.loc)[first:first + count] if with_loc else None, count))
            if keep_tiles:
                rows_slide += [s.name] * count
                rows_true += [s.y_true] * count
                if s.loc is not None:
                    rows_loc.append(np.asarray(s.loc)[first:first + count])

        for li, si, first, count, loaded, ev, is_rows in _feed_chunks(slides, mine, dev, copy_stream):
            if is_rows == 'z':                       # a compressed chunk: inflate on the decode CUs, un-filter here
                segs, cap = li, si
                t = decode_z(loaded, cap, count, ev, slides[segs[0][1]].source.tile_px, segs)
                assert t.shape[0] == count == sum(c for *_, c in segs)
                pend_tiles.append(t)
                for (li, si, first, c) in segs:
                    push(li, si, first, c, None)
                flush()
                continue
            if ev is not None:                       # made on the copy stream: order it before this stream's work
                cur_stream().wait_event(ev)
                loaded.record_stream(cur_stream())
                t = engines[0].png_unfilter(loaded) if is_rows else loaded.contiguous()
            elif isinstance(loaded, PngRows):        # filtered PNG scanlines: H2D, then the filters are reversed on the device
                t = engines[0].png_unfilter(_to_device(loaded.rows, dev))
            else:
                t = _to_device(loaded, dev)
            push(li, si, first, count, t)
            flush()
        flush(final=True)

    try:
        with (torch.cuda.stream(prep) if prep is not None else contextlib.nullcontext()):
            stream_slides()
    except BaseException:
        if table is not None:
            table.abort()
        raise
    if prep is not None:
        torch.cuda.current_stream(dev).wait_stream(prep)

    if pool:
        pool.synchronize()
    if monitor:
        try:
            headroom_look(block=True)
        except BaseException:
            if table is not None:
                table.abort()
            raise
    try:
        check_z(block=True)                         # what is left: the last chunks
    except BaseException:
        if table is not None:
            table.abort()
        raise
    live = [a for a in acc if a is not None]
    if live:
        # per-stream fixed-point accumulators are integers: their sum is exact and order-free
        tot = live[0] if len(live) == 1 else tuple(sum(a[j] for a in live[1:]) + live[0][j] for j in range(3))
        mp, mu, cnt = engines[0].slide_finish(tot)
        mp, mu, cnt = mp.cpu().numpy(), mu.cpu().numpy(), cnt.cpu().numpy()
    else:
        mp = mu = np.zeros(0); cnt = np.zeros(0, dtype=np.int64)
    # this rank's rows are on disk and its file closed BEFORE the collective: whoever leaves the gather knows every shard is complete
    table_path, table_rows = None, 0
    if table is not None:
        table_rows = table.finish()
        table_path = table.path
    cap = max(len(p) for p in parts) if parts else 0
    g_pred, g_unc, g_cnt = D.gather_slide_results(mine, mp[:n_local], mu[:n_local], cnt[:n_local],
                                                  len(slides), cap)
    if table is not None and world > 1 and rank == 0:
        table_path = assemble_shards(save_dir, table_name)
    tile_df = None
    if keep_tiles:
        mean = torch.cat(rows_mean).cpu().numpy() if rows_mean else np.zeros((0, 2), np.float32)
        std = torch.cat(rows_std).cpu().numpy() if rows_std else np.zeros((0, 2), np.float32)
        loc = np.concatenate(rows_loc) if rows_loc and sum(len(x) for x in rows_loc) == len(rows_slide) else None
        tile_df = tile_frame(outcome, rows_slide, rows_true, mean, std, loc if (with_loc or table is None) else None)
        if save_dir is not None and table is None:
            table_path = save_tile_predictions(tile_df, save_dir, table_name if world == 1 else shard_name(table_name, rank))
            table_rows = len(tile_df)
            if world > 1:                        # (pandas shards carry the same index, without byte ranges: row counts order them)
                order, at = [], 0
                for si in mine:
                    if slides[si].n_tiles:
                        order.append([si, slides[si].name, slides[si].n_tiles, 0, 0])
                write_shard_index(table_path, rank, world, outcome, 'loc_x' in tile_df.columns, order)
    return EvalResult(tile_df, [s.name for s in slides], g_pred, g_unc, g_cnt,
                      np.array([s.y_true for s in slides]), list(mine), table_path, table_rows, hr_state['min'], hr_state['checks'])
