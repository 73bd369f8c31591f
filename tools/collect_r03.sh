#!/bin/bash
# copy the round-3 evidence from gpurun_out/ (scratch) into profiles/ (tracked) and refresh the summaries
set -e
python tools/summarize_prof.py gpurun_out/prof_r03 profiles/r03_rocprof.md f16 > /dev/null
cp gpurun_out/prof_r03/trace/runc/*_kernel_stats.csv profiles/r03_kernel_stats.csv
cp gpurun_out/prof_r03/trace1/runc/*_kernel_stats.csv profiles/r03_kernel_stats_1stream.csv
cp gpurun_out/r03_sweep_mc.jsonl profiles/
python - <<'PY'
import json
for f in ('r03_bench', 'r03_bench_cfg3_share'):
    j = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    json.dump(j, open(f'profiles/{f}.json', 'w'), indent=1)
    print(f, round(j['value']), round(j['ms_per_step'], 3), (j.get('roofline') or {}).get('frac'), (j.get('roofline') or {}).get('traffic'))
PY
grep "Dominant" profiles/r03_rocprof.md
