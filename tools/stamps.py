"""Summarise s_memtime stamps written by BQ_STAMPS=<file> (kernels_mid.hip diagnostic build)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 8, 128).astype(np.int64)
nwg = int((a[:, 0, 0] > 0).sum())
print('workgroups with stamps:', nwg)
names = {0: 'start', 1: 'prologue issued', 2: 'barrier0', 3: 'D(0)', 4: 'barrier1', 56: 'residual dma+bar', 57: 'acc->lds', 58: 'barrier', 59: 'lds->global'}
for wg in (0, 1, min(9, nwg - 1)):
    for wave in (0, 4):
        t = a[wg, wave]
        t0 = t[0]
        print(f'-- wg {wg} wave {wave}')
        prev = t0
        for ev in [1, 2, 3, 4]:
            print(f'   {names[ev]:18s} +{t[ev]-prev:7d}  (@{t[ev]-t0})')
            prev = t[ev]
        for c in range(12):
            e = [t[5 + 4 * c + k] for k in range(4)]
            print(f'   chunk {c:2d}: top/D1 {e[0]-prev:6d}  G {e[1]-e[0]:6d}  D2 {e[2]-e[1]:6d}  barrier {e[3]-e[2]:6d}   (@{e[3]-t0})')
            prev = e[3]
        for ev in ([56] if t[56] > 0 else []) + [57, 58, 59]:
            print(f'   {names[ev]:18s} +{t[ev]-prev:7d}  (@{t[ev]-t0})')
            prev = t[ev]
# averages over workgroups and waves
t = a[:nwg]
t0 = t[:, :, 0:1]
def avg(x): return float(np.mean(x))
print('avg total cycles per tile:', avg(t[:, :, 59] - t[:, :, 0]))
print('avg prologue (start->barrier1):', avg(t[:, :, 4] - t[:, :, 0]))
loop = t[:, :, 8 + 4 * 11] - t[:, :, 4]
print('avg K loop:', avg(loop), ' per chunk', avg(loop) / 12)
print('avg epilogue:', avg(t[:, :, 59] - t[:, :, 8 + 4 * 11]))
for half, sl in (('first half (D then G)', slice(0, 4)), ('second half (G then D)', slice(4, 8))):
    th = t[:, sl]
    d1 = np.mean([np.mean(th[:, :, 5 + 4 * c] - (th[:, :, 4 + 4 * c] if c else th[:, :, 4])) for c in range(11)])
    g = np.mean([np.mean(th[:, :, 6 + 4 * c] - th[:, :, 5 + 4 * c]) for c in range(11)])
    d2 = np.mean([np.mean(th[:, :, 7 + 4 * c] - th[:, :, 6 + 4 * c]) for c in range(11)])
    b = np.mean([np.mean(th[:, :, 8 + 4 * c] - th[:, :, 7 + 4 * c]) for c in range(11)])
    print(f'{half}: store+D1 {d1:.0f}  G {g:.0f}  D2 {d2:.0f}  barrier wait {b:.0f}')

print('D stage of chunk 6 (issued in iteration 5): cycles between stamps 64..77')
for wave in (0, 4):
    d = np.diff(a[:nwg, wave, 64:78], axis=1)
    print(' wave', wave, 'mean', np.round(d.mean(0)).astype(int).tolist(), 'total', int((a[:nwg, wave, 77] - a[:nwg, wave, 64]).mean()))

e = a[:nwg]
print('epilogue: loop end->60 %d, residual_to_lds %d, barrier %d, acc_to_lds(+next loads) %d, barrier %d, lds_to_global %d' % (
    (e[:, :, 60] - e[:, :, 52]).mean(), (e[:, :, 61] - e[:, :, 60]).mean(), (e[:, :, 62] - e[:, :, 61]).mean(),
    (e[:, :, 57] - e[:, :, 62]).mean(), (e[:, :, 58] - e[:, :, 57]).mean(), (e[:, :, 59] - e[:, :, 58]).mean()))
