#!/bin/bash
# persistent-grid sizing under CU-masked streams (experiments build): two batches in flight, each on half the chip, with the
# kernels' grids sized for 256 CUs (default) against 128
cd ${GRAFT_REPO_ROOT:-.}
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 2>&1 | grep -E "error" | head -3
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), 'in flight', d['config']['hip_streams'])"; }
for i in 1 2; do
  python bench.py --steps 100 --streams 2 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | show "default  "
  BQ_NUM_CUS=128 python bench.py --steps 100 --streams 2 --no-extras --no-cpu-baseline --no-profile 2>/dev/null | show "numcus128"
done
