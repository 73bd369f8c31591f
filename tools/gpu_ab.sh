#!/bin/bash
# A/B on one box: the committed tree (_old, a git worktree built here) against the working tree, alternating
for r in 1 2 3; do
  for d in _old .; do
    ( cd $d && timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --streams 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=[k for k in d['kernels'] if '728_n728_19' in k['name']][0]
print('$d', 'step', round(d['ms_per_step'],3), '19x19', round(k['ms_per_launch'],4))
" )
  done
done
