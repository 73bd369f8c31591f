"""Condense rocprofv3 CSV output (tools/profile.sh) into a small markdown summary that is
committed under profiles/.  usage: python tools/summarize_prof.py gpurun_out/prof_r01 profiles/r01_rocprof.md [dtype]
(dtype = the --dtype the profiled bench ran with, default f16: the key of the figure in profiles/traffic.json)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+)I(.*)EEv(?:NS_)?\d*[A-Za-z]*Params', name)
    if m:        # mangled template arguments -> readable: DF16b = __bf16, DF16_ = _Float16, Li<n>E ints, Lb<0|1>E bools
        args = m.group(2).replace('DF16b', 'bf16,').replace('DF16_', 'f16,').replace('Li', '').replace('E', ',').replace('Lb', 'b')
        return f'{m.group(1)}<{args.strip(",")}>'
    name = name.replace('__bf16', 'bf16').replace('_Float16', 'f16')
    return name[:90]


def main():
    src, out = sys.argv[1], sys.argv[2]
    dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16'
    lines = [f'# rocprofv3 summary ({os.path.basename(src)})', '',
             'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 8 '
             '--warmup 2 --no-cpu-baseline --no-profile --no-extras` (the default: up to 4 batches in flight), the same with '
             '`--streams 1`, plus two separate `--pmc` passes, FETCH_SIZE and WRITE_SIZE, which do not fit one pass on gfx950.', '']
    def newest(pattern):
        f = sorted(glob.glob(pattern), key=os.path.getmtime)
        return f[-1:] if f else []

    stats = newest(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        lines += ['## Kernel time (--kernel-trace --stats)', '',
                  '| kernel | calls | total ms | avg us | % |', '|---|---:|---:|---:|---:|']
        for r in rows[:18]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                         f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
        lines += ['', f'Total GPU kernel time in trace: {tot/1e6:.1f} ms', '']
    trace = newest(os.path.join(src, 'trace', '*', '*_kernel_trace.csv'))
    if trace:
        rows = list(csv.DictReader(open(trace[0])))
        iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
        tot = sum(b - a for a, b in iv)
        uni, (cs, ce) = 0, iv[0]
        for a, b in iv[1:]:
            if a > ce:
                uni += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        uni += ce - cs
        queues = len({r['Queue_Id'] for r in rows})
        lines += [f'Default run (several batches in flight on CU-masked streams): {queues} hardware queue(s); sum of kernel durations {tot/1e6:.1f} ms, union of their '
                  f'intervals {uni/1e6:.1f} ms: kernels of the batches in flight run side by side on disjoint XCD groups, so a launch\'s '
                  'start..end interval in this trace is its duration on a fraction of the chip.', '']
    trace1 = newest(os.path.join(src, 'trace1', '*', '*_kernel_trace.csv'))
    per_launch = trace1 or trace
    if per_launch:
        # the same kernel symbol serves several layers: split by grid size so a layer's average
        # launch duration can be compared with bench.py's event-timed figure
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(per_launch[0])):
            k = (short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))
            acc[k][0] += 1
            acc[k][1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        which = ('the same command with `--streams 1` (no overlap between launches; this is what bench.py\'s '
                 'per-launch HIP-event durations are comparable with)') if trace1 else 'the two-stream trace'
        lines += ['## Per (kernel, workgroups) average launch duration', '', f'From {which}.', '',
                  '| kernel | workgroups | launches | avg us | total ms |', '|---|---:|---:|---:|---:|']
        for (kn, wg), (n, dur) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
            lines.append(f'| `{kn}` | {wg} | {n} | {dur/n/1e3:.1f} | {dur/1e6:.2f} |')
        # the 728 -> 728 @19x19 layer class = the Geo<19,4> instances with K = 736 (persistent: one workgroup per CU)
        dom = [(n, dur) for (kn, wg), (n, dur) in acc.items() if 'sepconv_wide_kernel' in kn and 'GeoI19' in kn and ',,,736,736,736,' in kn]
        if dom:
            n = sum(x[0] for x in dom); dur = sum(x[1] for x in dom)
            lines += ['', f'Dominant layer class (sepconv 728->728 @19x19, kernels_wide.hip, all its instances: with / without a '
                          f'residual input, with / without a ReLU in front): {n} launches, average {dur/n/1e3:.1f} us.']
        lines.append('')
    pmc = {}
    for key in ('pmc_fetch', 'pmc_write'):
        files = newest(os.path.join(src, key, '*', '*_counter_collection.csv'))
        if not files:
            continue
        acc = defaultdict(lambda: [0.0, 0, 0.0])
        for r in csv.DictReader(open(files[0])):
            wgs = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
            a = acc[(short(r['Kernel_Name']) + f' [{wgs} wg]', r['Counter_Name'])]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
            a[2] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        pmc[key] = acc
    if pmc:
        names = defaultdict(dict)
        for key, acc in pmc.items():
            for (kn, cn), (v, n, dur) in acc.items():
                names[kn][cn] = (v / n, n, dur / n)
        lines += ['## HBM traffic per launch (--pmc FETCH_SIZE / WRITE_SIZE, separate passes)', '',
                  'Counter units are KiB.  Corrected bytes follow MI355X_MICROARCH.md (HBM section): on gfx950 '
                  'FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are doubled; WRITE_SIZE is exact.',
                  '', '| kernel | launches | FETCH_SIZE avg (KiB) | WRITE_SIZE avg (KiB) | corrected MB/launch | avg us (pmc pass) |',
                  '|---|---:|---:|---:|---:|---:|']
        order = sorted(names.items(), key=lambda kv: -sum(x[0] * x[1] for x in kv[1].values()))
        for kn, d in order[:14]:
            f = d.get('FETCH_SIZE', (0, 0, 0)); w = d.get('WRITE_SIZE', (0, 0, 0))
            mb = (2 * f[0] + w[0]) * 1024 / 1e6
            lines.append(f'| `{kn}` | {f[1] or w[1]} | {f[0]:.0f} | {w[0]:.0f} | {mb:.1f} | {(f[2] or w[2])/1e3:.1f} |')
        lines.append('')
        # the dominant kernel of bench.py (728 -> 728 separable conv at 19x19, n = 256 -> 963 workgroups)
        dom = [(kn, d) for kn, d in names.items() if 'sepconv_wide_kernel' in kn and 'GeoI19' in kn and ',,,736,736,736,' in kn]
        if dom:
            import json
            nl = sum((d.get('FETCH_SIZE') or d.get('WRITE_SIZE'))[1] for _, d in dom)
            fetch = sum(d['FETCH_SIZE'][0] * d['FETCH_SIZE'][1] for _, d in dom if 'FETCH_SIZE' in d) / max(nl, 1)
            write = sum(d['WRITE_SIZE'][0] * d['WRITE_SIZE'][1] for _, d in dom if 'WRITE_SIZE' in d) / max(nl, 1)
            tpath = os.path.join(os.path.dirname(out), 'traffic.json')
            try:
                js = json.load(open(tpath))
                if '_source' not in js:          # round-2 layout (one bf16 entry at top level)
                    js = {'bf16': js}
            except (OSError, ValueError):
                js = {}
            js['_source'] = ('rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over '
                             '`bench.py --steps 8 --warmup 2 --dtype <key>` at batch 256 on one MI355X (tools/profile.sh); '
                             'corrected = (2 x FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md; per-dtype `source` names '
                             'the committed summary')
            js.setdefault(dtype, {})['sepconv_k728_n728_19x19'] = {
                'fetch_size_kib': fetch, 'write_size_kib': write, 'corrected_bytes_per_launch': (2 * fetch + write) * 1024,
                'source': os.path.basename(out), 'launches': nl}
            json.dump(js, open(tpath, 'w'), indent=1)
            lines += [f'Dominant kernel (728 -> 728 @19x19, all instances): FETCH_SIZE {fetch:.0f} KiB, WRITE_SIZE {write:.0f} KiB per launch '
                      f'-> corrected {(2 * fetch + write) * 1024 / 1e6:.1f} MB (algorithmic 270 MB for the 17 launches without and 406 MB for the 8 with a residual input: 314 MB class-weighted).', '']
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:40]))


if __name__ == '__main__':
    main()
