#!/bin/bash
# Does the middle flow run faster when its tensors fit the 256 MB Infinity Cache?  Batch sizes whose 19x19 tile counts divide over
# 256 persistent workgroups (153 -> 765 tiles = 3 each, 205 -> 1 025 = 4 each, 256 -> 1 280 = 5 each), per-tile time of the dominant kernel.
cd ${GRAFT_REPO_ROOT:-.}
for b in 256 205 153 102 256; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 --batch $b --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['name']:k for k in d['kernels']}
w=ks['sepconv_k728_n728_19x19']
print('batch $b', 'tiles/s %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'us per tile-image %.3f' % (d['ms_per_step']*1e3/$b),
      '19x19: %.4f ms/launch = %.4f us per image' % (w['ms_per_launch'], w['ms_per_launch']*1e3/$b), 'entry_side %.3f' % d['entry_side_ms'])
"
done
