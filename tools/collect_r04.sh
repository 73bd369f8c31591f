#!/bin/bash
# copy the round-4 evidence from gpurun_out/ (scratch) into profiles/ (tracked) and refresh the summaries
set -e
python tools/summarize_prof.py gpurun_out/prof_r04 profiles/r04_rocprof.md f16 > /dev/null
cp gpurun_out/prof_r04/trace/runc/*_kernel_stats.csv profiles/r04_kernel_stats.csv
cp gpurun_out/prof_r04/trace1/runc/*_kernel_stats.csv profiles/r04_kernel_stats_1stream.csv
cp gpurun_out/r04_sweep_mc.jsonl profiles/
python - <<'PY'
import json
for f in ('r04_bench', 'r04_bench_cfg3_share'):
    j = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    json.dump(j, open(f'profiles/{f}.json', 'w'), indent=1)
    print(f, round(j['value']), round(j['ms_per_step'], 3), (j.get('roofline') or {}).get('frac'), (j.get('roofline') or {}).get('traffic'))
PY
grep "Dominant" profiles/r04_rocprof.md
