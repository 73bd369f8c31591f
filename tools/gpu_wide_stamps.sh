#!/bin/bash
# experiments build on the box, steady-state stamps (5th round of tiles) of the wide kernel + per-kernel time
mkdir -p gpurun_out
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 EXPFLAGS=-DWIDE_ABLATE=${1:-0} 2>&1 | grep -E "error" | head -3
BQ_STAMPS_B0=${2:-64} BQ_STAMPS_WIDE=gpurun_out/stamps_wide.bin timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 4 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in d['kernels']:
    if '728_n728_19' in k['name']: print('   ', k['name'], round(k['ms_per_launch'],4))
"
python tools/stamps_wide.py gpurun_out/stamps_wide.bin
