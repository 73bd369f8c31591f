#!/bin/bash
# config 3's per-GPU share through evaluate() next to the config-2 loop, twice each, on one box
cd ${GRAFT_REPO_ROOT:-.}
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), 'in flight', d['config']['hip_streams'])"; }
for i in 1 2; do
  python bench.py --workload cfg3 --gpus 1 --slides 200 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | show cfg3
  python bench.py --steps 100 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | show cfg2
done
