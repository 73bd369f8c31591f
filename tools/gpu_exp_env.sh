#!/bin/bash
# experiments build on the box, then the per-kernel table with the given environment (e.g. BQ_TILE_MASK=15)
mkdir -p gpurun_out
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 2>&1 | grep -E "error" | head -3
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --streams 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('step', round(d['ms_per_step'],3))
for k in d['kernels'][:12]: print(f\"  {k['name']:34s} {k['ms_per_launch']:.4f}\")
"
done
