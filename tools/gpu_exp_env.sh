#!/bin/bash
# experiments build on the box, then the per-kernel table with each given environment (e.g. BQ_TILE_MASK=15);
# KERNELS="gemm_ split_" restricts the table to kernels whose names contain one of the words
mkdir -p gpurun_out
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 2>&1 | grep -E "error" | head -3
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --streams 1 2>/dev/null | tail -1 | python tools/bench_kernels.py $KERNELS
done
