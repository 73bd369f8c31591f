#!/bin/bash
# A/B on one box: the wide kernel built with and without an extra compile flag ($1, e.g. -DWIDE_PRIO=1); per-kernel ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for flags in "" "$1"; do
  rm -f biscuit_amd/csrc/build/kernels_wide.o biscuit_amd/csrc/build/wide.ok
  make -C biscuit_amd/csrc -j8 EXPFLAGS="$flags" EXPERIMENTS=1 2>&1 | grep -E " error" | head -3
  echo "== flags: '$flags'"
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --streams 1 2>/dev/null | tail -1 | python tools/bench_kernels.py 728_n728 256_n728
done
done
