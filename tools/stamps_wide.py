"""Decode BQ_STAMPS_WIDE dumps (experiments build): per-phase cycles of kernels_wide.hip, median over blocks and waves."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(64, 8, 32).astype(np.int64)
ok = a[:, :, 0] > 0
names = ['const', 'issue', 'wait0', 'bar0', 'D(0)', 'wait1+bar'] + [f'chunk{c}' for c in range(12)] + ['(gap)', 'res-issue', 'res-wait', 'crumbs', 'rows-out']
ev = list(range(0, 18)) + [19, 20, 21, 22]
rows = []
for b in range(64):
    for w in range(8):
        if ok[b, w]:
            t = a[b, w]
            seq = [t[e] for e in ev if t[e] > 0]
            rows.append((t, seq))
print('blocks*waves with stamps:', len(rows))
t = np.array([r[0] for r in rows])
def med(x): return int(np.median(x))
print('total', med(t[:, 22] - t[:, 0]))
prev = 0
for e in range(1, 23):
    if (t[:, e] > 0).all():
        print(f'  ev{e:2d} +{med(t[:, e] - t[:, prev]):7d}   (since start {med(t[:, e] - t[:, 0])})')
        prev = e
