#!/bin/bash
# same-box A/B of kernels_front.hip variants: build the library with each EXTRA flag set in turn and time the front kernel
cd ${GRAFT_REPO_ROOT:-.}
for v in "" "-DFRONT_NW=12" "-DFRONT_NW=16" "-DFRONT_NW=8" ""; do
  touch biscuit_amd/csrc/kernels_front.hip
  make -C biscuit_amd/csrc EXTRA="$v" -j4 >/dev/null 2>&1
  echo "variant [$v]"
  python bench.py --steps 30 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python tools/bench_kernels.py front step
done
