"""What runs NEXT to a kernel decides how long it takes?  From a rocprofv3 --kernel-trace CSV of a run with two batches in flight
on disjoint halves of the chip: every launch of the 728 -> 728 @19x19 kernel, classified by what the OTHER stream ran during it
(another 19x19 launch, an entry-side kernel, nothing), with its duration.  usage: python tools/pairing.py <kernel_trace.csv> [label]
"""
import sys

import numpy as np
import pandas as pd


def klass(name):
    if 'sepconv_wide_kernel' in name:
        return 'wide19' if 'GeoILi19' in name else 'wide37_74'
    if any(k in name for k in ('front_stream', 'sepconv_stream', 'block_tail', 'stage_stats')):
        return 'entry'
    if 'gemm_tile' in name:
        return 'poolgemm'
    if any(k in name for k in ('exit_gemm', 'dw3x3', 'head_')):
        return 'exit_head'
    return 'other'


def main():
    df = pd.read_csv(sys.argv[1])
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    df = df.sort_values('Start_Timestamp').reset_index(drop=True)
    df['cls'] = df['Kernel_Name'].map(klass)
    df['dur'] = (df['End_Timestamp'] - df['Start_Timestamp']) / 1e3            # us
    t0 = df['Start_Timestamp'].min() + 0.3 * (df['End_Timestamp'].max() - df['Start_Timestamp'].min())   # skip warm-up / calibration
    st, en, cl, q = (df[c].to_numpy() for c in ('Start_Timestamp', 'End_Timestamp', 'cls', 'Queue_Id'))
    rows = []
    for i in np.flatnonzero((cl == 'wide19') & (st >= t0)):
        lo = np.searchsorted(st, st[i] - 5_000_000)
        ov = {}
        for j in range(lo, len(st)):
            if st[j] >= en[i]:
                break
            if j == i or q[j] == q[i]:
                continue
            o = min(en[i], en[j]) - max(st[i], st[j])
            if o > 0:
                ov[cl[j]] = ov.get(cl[j], 0) + o
        d = en[i] - st[i]
        tot = sum(ov.values())
        if tot < 0.1 * d:
            pair = 'alone'
        else:
            top = max(ov, key=ov.get)
            pair = top if ov[top] >= 0.6 * d else 'mixed'
        res = 'res' if 'Lb0ELb1E' in df['Kernel_Name'][i] or 'Lb1ELb1E' in df['Kernel_Name'][i] else 'plain'
        rows.append((res, pair, d / 1e3))
    r = pd.DataFrame(rows, columns=['variant', 'next_to', 'us'])
    print(f'== {label}: {len(r)} launches of the 19x19 kernel; step kernels on {df["Queue_Id"].nunique()} queues')
    if len(r):
        print(r.groupby(['variant', 'next_to'])['us'].agg(['count', 'mean', 'median', 'min']).round(1).to_string())
    ent = df[(df['cls'] == 'entry') & (df['Start_Timestamp'] >= t0)]
    if len(ent):
        g = ent.groupby(ent['Kernel_Name'].str.slice(18, 60))['dur'].agg(['count', 'mean']).round(1)
        print(g.to_string())


if __name__ == '__main__':
    main()
