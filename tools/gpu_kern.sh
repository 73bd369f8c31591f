#!/bin/bash
# Per-kernel table of one default-size bench run (HIP events per launch, one stream) + the headline: bash tools/gpu_kern.sh TAG [n]
# -> gpurun_out/TAG_kern.txt.  The A/B harness of round 6's kernel campaign: run it on the same box for both builds.
T=${1:-x}; N=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for i in $(seq 1 $N); do
python bench.py --no-extras --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value %.0f tiles/s  ms_per_step %.4f  entry_side_ms %.4f  streams %s' % (j['value'], j['ms_per_step'], j.get('entry_side_ms',0), j['config']['hip_streams']))
for k in j['kernels'][:22]: print('  %-44s x%-4.0f %.4f ms  frac %.3f' % (k['name'], k['launches_per_step'], k['ms_per_launch'], k['frac_of_bound']))
"
done | tee gpurun_out/${T}_kern.txt
