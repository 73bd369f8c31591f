#!/bin/bash
# per-kernel table of bench.py for the two 16-bit types on one box (one stream each, then the default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for dt in f16 bf16; do
  python3 $R/bench.py --dtype $dt --steps 40 --warmup 3 --no-extras --no-cpu-baseline --streams 1 > $R/gpurun_out/ab_${dt}_1s.json 2> $R/gpurun_out/ab_${dt}_1s.err
done
python3 - <<'PY'
import json, os
R=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
a={}
for dt in ('f16','bf16'):
    try:
        j=json.loads([l for l in open(f'{R}/gpurun_out/ab_{dt}_1s.json') if l.startswith('{')][-1])
    except Exception as e:
        print(dt, 'failed', e, open(f'{R}/gpurun_out/ab_{dt}_1s.err').read()[-2000:]); continue
    a[dt]=j
    print(dt, 'value', round(j['value']), 'ms/step', round(j['ms_per_step'],3), 'roofline', j['roofline']['kernel'], round(j['roofline']['avg_launch_ms'],4), round(j['roofline']['frac'],4), 'sum(share)', round(sum(k['share'] for k in j['kernels']),3))
if len(a)==2:
    kb={k['name']:k for k in a['bf16']['kernels']}
    print(f"{'kernel':44s} {'n':>4s} {'f16 ms':>8s} {'bf16 ms':>8s}  ratio")
    for k in a['f16']['kernels']:
        b=kb.get(k['name'])
        if b: print(f"{k['name']:44s} {k['launches_per_step']:4.0f} {k['ms_per_launch']:8.4f} {b['ms_per_launch']:8.4f}  {k['ms_per_launch']/b['ms_per_launch']:.3f}")
PY
