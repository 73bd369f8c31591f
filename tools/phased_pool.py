"""EXPERIMENT, not product: two batches in flight with every batch cut into its ENTRY part and its REST (``bq_mc_infer_part``), the parts
of consecutive batches scheduled against each other on CU-masked streams.  Round 5 measured every such schedule equal to or slower
than ``EnginePool``'s free-running streams (profiles/r05_schedules_steps.log), so round 6 took it out of the package; it needs the
experiments build of the library (``make -C biscuit_amd/csrc EXPERIMENTS=1``), which still exports ``bq_mc_infer_part``."""
import ctypes as C

import torch

from biscuit_amd.engine import Engine, _mask_stream, _ptr

BQ_PART = {'entry': 1, 'rest': 2, 'all': 3}


def mc_infer_part(eng, part, tiles_u8, mc_n, seed, tile_idx0=0, out=None):
    """``Engine.mc_infer`` (head mode) in two parts: 'entry' then 'rest' with the same arguments on one engine equals it bit for bit."""
    fn = eng._lib.bq_mc_infer_part                       # (AttributeError on the product build: the symbol is not there)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_size_t, C.c_void_p]
    n = tiles_u8.shape[0]
    ws = eng._ws_for(n, mc_n)
    mean, std = out
    eng._check(fn(eng._ctx, _ptr(tiles_u8), n, int(tile_idx0), int(mc_n), int(seed), BQ_PART[part], _ptr(mean), _ptr(std), _ptr(ws),
                  ws.numel(), eng._stream()))
    return mean, std


class PhasedPool:
    """Two batches in flight, each cut into its two parts (``mc_infer_part`` above): ENTRY = staging, stem and entry flow --
    vector-ALU / HBM-bound kernels at 2.0-2.3 GHz --, REST = middle and exit flow + MC head -- matrix-core kernels that run
    against the chip's power management at 1.35-1.45 GHz.  ``EnginePool`` lets two streams drift; here the pairing is chosen:

    * ``schedule='antiphase'``: two streams on disjoint halves of the chip, batch i on stream i % 2; a batch's ENTRY part does
      not start before the previous batch's ENTRY part (other stream) has finished, so one half runs ENTRY while the other
      runs REST for all but |REST - ENTRY| of a period.
    * ``schedule='pipeline'``: one stream owns ``cus_entry`` compute units and runs every batch's ENTRY part, the other owns the
      rest of the chip and runs every REST part (uneven splits: the two parts are not equally long); batch i uses engine
      i % n_engines, whose workspace carries the entry flow's output from one stream to the other.
    Results are those of ``mc_infer``, bit for bit, whatever the schedule."""

    def __init__(self, weights, schedule='antiphase', cus_entry=None, n_engines=2, size_grids=True, **kw):
        if schedule not in ('antiphase', 'pipeline'):
            raise ValueError(schedule)
        self.schedule = schedule
        self.engines = [Engine(weights, **kw) for _ in range(max(2, int(n_engines)))]
        self.device = self.engines[0].device
        self.hp = self.engines[0].hp
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.ncu = ncu
        self.size_grids = bool(size_grids)
        e0 = self.engines[0]
        if schedule == 'antiphase':
            self.cus = (ncu // 2, ncu - ncu // 2)
            self.streams = [_mask_stream(e0, range(0, ncu // 2), ncu), _mask_stream(e0, range(ncu // 2, ncu), ncu)]
            if self.size_grids:
                for k, eng in enumerate(self.engines):
                    eng.set_num_cus(self.cus[k % 2])
        else:
            ne = int(cus_entry or ncu // 2)
            if not 8 <= ne <= ncu - 8:
                raise ValueError(f'cus_entry must lie in [8, {ncu - 8}]')
            self.cus = (ne, ncu - ne)
            self.streams = [_mask_stream(e0, range(0, ne), ncu), _mask_stream(e0, range(ne, ncu), ncu)]
        self._entry_done = None                         # antiphase: the previous batch's ENTRY part
        self._rest_done = [None] * len(self.engines)    # pipeline: the last REST part on each engine's workspace

    def __len__(self):
        return 2

    def step(self, i, tiles_u8, mc_n, seed, tile_idx0, out, after=None):
        """Enqueue batch i: (mean, std) -> ``out``; ``after(engine)`` runs behind the REST part on its stream (the slide reduce).
        ``tiles_u8`` and ``out`` must already be valid on the streams (resident inputs, pre-allocated outputs)."""
        k = i % len(self.engines)
        eng = self.engines[k]
        if self.schedule == 'antiphase':
            st = self.streams[k % 2]
            if self._entry_done is not None:
                st.wait_event(self._entry_done)
            with torch.cuda.stream(st):
                mc_infer_part(eng, 'entry', tiles_u8, mc_n, seed, tile_idx0=tile_idx0, out=out)
                ev = torch.cuda.Event()
                ev.record(st)
                self._entry_done = ev
                mc_infer_part(eng, 'rest', tiles_u8, mc_n, seed, tile_idx0=tile_idx0, out=out)
                if after is not None:
                    after(eng)
            return
        se, sr = self.streams
        if self._rest_done[k] is not None:
            se.wait_event(self._rest_done[k])            # the workspace is free again
        with torch.cuda.stream(se):
            if self.size_grids:
                eng.set_num_cus(self.cus[0])
            mc_infer_part(eng, 'entry', tiles_u8, mc_n, seed, tile_idx0=tile_idx0, out=out)
            ev = torch.cuda.Event()
            ev.record(se)
        sr.wait_event(ev)
        with torch.cuda.stream(sr):
            if self.size_grids:
                eng.set_num_cus(self.cus[1])
            mc_infer_part(eng, 'rest', tiles_u8, mc_n, seed, tile_idx0=tile_idx0, out=out)
            if after is not None:
                after(eng)
            ev2 = torch.cuda.Event()
            ev2.record(sr)
            self._rest_done[k] = ev2

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        self.synchronize()
        for st in self.streams:
            self.engines[0]._lib.bq_stream_destroy(self.engines[0]._ctx, C.c_void_p(st.cuda_stream))
        self.streams = []
