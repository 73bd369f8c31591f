#!/bin/bash
# experiments build on the box (the product .so travels; rebuild with EXPERIMENTS=1 into a side directory is avoided:
# build in place, it is a scratch copy), stamps of the wide kernel
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 2>&1 | grep -E "error|check_" | head
mkdir -p gpurun_out
BQ_STAMPS_WIDE=gpurun_out/stamps_wide.bin timeout 600 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 4 --streams 1 2>&1 | tail -1 | cut -c1-200
python tools/stamps_wide.py gpurun_out/stamps_wide.bin
BQ_STAMPS_NORES=1 BQ_STAMPS_WIDE=gpurun_out/stamps_wide_nores.bin timeout 600 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 4 --streams 1 2>&1 | tail -1 | cut -c1-100
python tools/stamps_wide.py gpurun_out/stamps_wide_nores.bin
