#!/bin/bash
# the per-kernel table for builds with extra compiler flags (one build per argument; "none" = the normal flags)
mkdir -p gpurun_out
for f in "$@"; do
  echo "== $f"
  make -C biscuit_amd/csrc clean >/dev/null 2>&1
  if [ "$f" = none ]; then EF=""; else EF="$f"; fi
  make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 EXPFLAGS="$EF" 2>&1 | grep -E "error|FAILED" | head -3
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --streams 1 2>/dev/null | tail -1 | python tools/bench_kernels.py $KERNELS
done
