"""Decode BQ_STAMPS_WIDE dumps (experiments build) of the persistent kernels_wide.hip: s_memtime stamps [64 workgroups][8 waves]
[8 tiles][32 events]; per-phase ticks, median over workgroups and waves, tile by tile.
events: 0-5 once per workgroup (constants, loads issued, wait, barrier, D(0), wait + barrier); per tile: 18 = tile start,
6+c = end of chunk c (after its closing barrier), 20 = epilogue start (accumulators readable), 22 = epilogue end."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 8, 8, 32).astype(np.int64)
ok = a[:, :, 0, 0] > 0
print('workgroups x waves with stamps:', int(ok.sum()))
t = a[ok]                       # [n, tile, ev]
med = lambda x: int(np.median(x))
print('prologue (once): issue %d, wait %d, barrier %d, D(0) %d, wait+barrier %d' % tuple(med(t[:, 0, e + 1] - t[:, 0, e]) for e in range(5)))
for it in range(8):
    if not (t[:, it, 22] > 0).all():
        break
    s = t[:, it]
    chunks = [med(s[:, 6] - s[:, 18])] + [med(s[:, 6 + c] - s[:, 5 + c]) for c in range(1, 12)]
    gap = med(s[:, 18] - t[:, it - 1, 22]) if it else med(s[:, 18] - s[:, 5])
    print(f'tile {it}: total {med(s[:, 22] - s[:, 18])}  gap before {gap}  chunks {chunks}  nops {med(s[:, 20] - s[:, 17])}  epilogue {med(s[:, 22] - s[:, 20])}')
    # spread over the waves of a workgroup: when does the last wave finish its epilogue after the first?
# in-kernel clock: cycles (s_memtime) per 10 ns tick of s_memrealtime, first tile start -> last stamped tile's end
last = max(it for it in range(8) if (t[:, it, 22] > 0).all())
cyc = t[:, last, 22] - t[:, 0, 18]
rt = t[:, last, 31] - t[:, 0, 30]
if (rt > 0).all():
    print(f'in-kernel clock over tiles 0..{last}: median {np.median(cyc / rt) * 0.1:.3f} GHz (min {np.min(cyc / rt) * 0.1:.3f}, max {np.max(cyc / rt) * 0.1:.3f})')
w = a[:, :, :, :]
blocks = [b for b in range(64) if (a[b, :, 0, 0] > 0).all()]
for it in range(2):
    sp = [a[b, :, it, 22].max() - a[b, :, it, 22].min() for b in blocks if (a[b, :, it, 22] > 0).all()]
    if sp:
        print(f'tile {it}: spread of the epilogue end over the 8 waves of a workgroup: median {int(np.median(sp))}')
if blocks:
    b = blocks[0]
    print('whole workgroup 0: first stamp -> last stamp', int(a[b][a[b] > 0].max() - a[b][a[b] > 0].min()))
