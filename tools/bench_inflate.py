"""Throughput of the device inflate (bq_png_inflate + un-filter) on the GPU box: tiles/s for nearly incompressible and photo-like PNG
tiles, as a function of the streams in flight and of the compute units it may use (a CU-masked stream), alone and beside a running
inference.  usage: python tools/bench_inflate.py [--n 4096] [--beside]"""
import argparse, io, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biscuit_amd import tfrecord as tfr, tfrecord_native as tn
from biscuit_amd.engine import Engine, EnginePool, _mask_stream
from biscuit_amd.synthetic import make_tiles
from biscuit_amd.weights import synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, nargs='+', default=[1024, 4096, 8192])
ap.add_argument('--cus', type=int, nargs='+', default=[256, 32, 16])
ap.add_argument('--variant', type=int, nargs='+', default=[0, 5])
ap.add_argument('--beside', action='store_true', help='also with a two-stream inference loop running on the rest of the chip')
args = ap.parse_args()
dev = torch.device('cuda', 0)
w = synthetic_weights(1)
eng = Engine(w, dtype='f16', max_batch=256, max_mc=30)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count

def streams_of(kind, n):
    grain = {'noise': 18.0, 'photo': 4.0}[kind]
    base = [tfr.encode_image(t) for t in make_tiles(32, seed=21, grain=grain)]
    d = tempfile.mkdtemp(prefix='bq_inf_')
    p = os.path.join(d, 's.tfrecords')
    tfr.write_slide(p, 's', [base[i % 32] for i in range(n)], np.zeros((n, 2), np.int64))
    with tn.NativeReader(p) as r:
        z = np.zeros(n * 240000, np.uint8)
        off, ln = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        t0 = time.perf_counter(); used, _ = r.extract_z(0, n, 299, z, off, ln); t_ex = time.perf_counter() - t0
        t0 = time.perf_counter(); used, _ = r.extract_z(0, n, 299, z, off, ln); t_ex = min(t_ex, time.perf_counter() - t0)
        want = r.decode(0, min(n, 64))[0]
    os.remove(p); os.rmdir(d)
    return z[:used], off, ln, want, t_ex

for kind in ('noise', 'photo'):
    for n in args.n:
        z, off, ln, want, t_ex = streams_of(kind, n)
        zd, od, ld = torch.from_numpy(z).to(dev), torch.from_numpy(off.view(np.int32)).to(dev), torch.from_numpy(ln.view(np.int32)).to(dev)
        tiles, status = eng.png_decode_z(zd, od, ld)
        torch.cuda.synchronize()
        assert not status.cpu().numpy().any() and np.array_equal(tiles[:want.shape[0]].cpu().numpy(), want)
        print(f'{kind} n={n}: {z.size / n / 1e3:.0f} KB compressed per tile; host extract_z {n / t_ex:.0f} tiles/s ({len(os.sched_getaffinity(0))} cores)', flush=True)
        for variant, cus in [(v, c) for v in args.variant for c in args.cus]:
            eng.set_option('inflate_variant', variant)
            st = torch.cuda.Stream(device=dev) if cus >= ncu else _mask_stream(eng, range(ncu - cus, ncu), ncu)
            with torch.cuda.stream(st):
                eng.png_decode_z(zd, od, ld)
                a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                a.record(st)
                rows, status = eng.png_inflate(zd, od, ld)
                b.record(st)
                eng.png_decode_z(zd, od, ld)
                c.record(st)
            st.synchronize()
            t_inf, t_all = a.elapsed_time(b) * 1e-3, b.elapsed_time(c) * 1e-3
            print(f'    variant {variant} {cus:3d} CUs: inflate {n / t_inf:8.0f} tiles/s ({t_inf * 1e3:.1f} ms);  inflate + un-filter {n / t_all:8.0f} tiles/s', flush=True)
        if args.beside:
            eng.set_option('inflate_variant', args.variant[0])
            # inference on CUs [0, ncu - 16) in two halves, the decoder on the last 16: step time with and without the decoder running
            pool = EnginePool(w, n_streams=1, dtype='f16', max_batch=256, max_mc=30)
            inf_st = _mask_stream(pool.engines[0], range(0, ncu - 16), ncu)
            dec_st = _mask_stream(eng, range(ncu - 16, ncu), ncu)
            batch = torch.randint(0, 256, (256, 299, 299, 3), dtype=torch.uint8, device=dev)
            def infer(k):
                with torch.cuda.stream(inf_st):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(inf_st)
                    for _ in range(k):
                        pool.engines[0].mc_infer(batch, 30, 1)
                    b.record(inf_st)
                return a, b
            a, b = infer(10); inf_st.synchronize()
            a, b = infer(20); inf_st.synchronize(); alone = a.elapsed_time(b) / 20
            with torch.cuda.stream(dec_st):
                c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c.record(dec_st)
                reps = 0
                for _ in range(max(1, int(0.8 / (n / 17000.0)))):          # ~0.8 s of decoding: longer than the 20 inference steps below
                    eng.png_decode_z(zd, od, ld); reps += 1
                d.record(dec_st)
            a, b = infer(20)
            inf_st.synchronize(); dec_st.synchronize()
            print(f'    beside inference (240 + 16 CUs): step {alone:.3f} ms alone -> {a.elapsed_time(b) / 20:.3f} ms with the decoder running; '
                  f'decoder {reps * n / (c.elapsed_time(d) * 1e-3):.0f} tiles/s', flush=True)
            pool.close()
