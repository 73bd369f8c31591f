#!/bin/bash
# same-box A/B of a build flag that only touches csrc/kernels_exit.hip: bash tools/gpu_exit_ab.sh -DFLAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "" "$@"; do
  touch biscuit_amd/csrc/kernels_exit.hip
  make -C biscuit_amd/csrc EXTRA="$v" 2>&1 | grep -E "error" | head -3
  timeout 600 python bench.py --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', 'step', round(d['ms_per_step'],3), ' '.join(k['name'][:22]+' '+str(round(k['ms_per_launch'],4)) for k in d['kernels'] if k['name'].startswith(('gemm_', 'respool_19'))))
"
done
