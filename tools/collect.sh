#!/bin/bash
# copy a round's evidence from gpurun_out/ (scratch) into profiles/ (tracked) and refresh the summaries: bash tools/collect.sh r05
T=${1:-r06}
set -e
python tools/summarize_prof.py gpurun_out/prof_${T} profiles/${T}_rocprof.md f16 > /dev/null
cp "$(ls -t gpurun_out/prof_${T}/trace/runc/*_kernel_stats.csv | head -1)" profiles/${T}_kernel_stats.csv      # (gpurun_out accumulates: the newest run's)
cp "$(ls -t gpurun_out/prof_${T}/trace1/runc/*_kernel_stats.csv | head -1)" profiles/${T}_kernel_stats_1stream.csv
cp gpurun_out/${T}_sweep_mc.jsonl profiles/
TAG=$T python - <<'PY'
import json, os
T = os.environ['TAG']
for f in (f'{T}_bench', f'{T}_bench_cfg3_share'):
    j = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][-1])
    json.dump(j, open(f'profiles/{f}.json', 'w'), indent=1)
    print(f, round(j['value']), round(j['ms_per_step'], 3), (j.get('roofline') or {}).get('frac'), (j.get('roofline') or {}).get('traffic'))
PY
grep "Dominant" profiles/${T}_rocprof.md
