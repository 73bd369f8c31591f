#!/bin/bash
# experiments builds of the wide kernel with parts removed (timing only), steady-state stamps (5th round of tiles)
mkdir -p gpurun_out
for ab in ${@:-0 1 2 3 4}; do
  make -C biscuit_amd/csrc clean >/dev/null 2>&1
  make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 EXPFLAGS=-DWIDE_ABLATE=$ab 2>&1 | grep -E "error" | head -3
  echo "== WIDE_ABLATE=$ab (1: no B reload, 2: no depthwise/convert, 4: no MFMA, 8: no convert, 16: no A-fragment reads)"
  BQ_STAMPS_B0=64 BQ_STAMPS_WIDE=gpurun_out/stamps_ab$ab.bin timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 4 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in d['kernels']:
    if '728_n728_19' in k['name']: print('   ', k['name'], round(k['ms_per_launch'],4))
"
  python tools/stamps_wide.py gpurun_out/stamps_ab$ab.bin | grep -E "tile 3|tile 4|in-kernel clock"
done
