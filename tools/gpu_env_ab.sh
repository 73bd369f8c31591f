#!/bin/bash
# experiments build on the box; per-kernel table with and without one environment switch ($1, e.g. BQ_TILE_SEPQ=1), twice;
# the remaining arguments restrict the table to kernels whose names contain one of them
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
SW=$1; shift
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 2>&1 | grep -E " error" | head -3
python tools/act_hash.py 5 2>/dev/null | grep -v amdgpu > gpurun_out/hash_off.log
env $SW python tools/act_hash.py 5 2>/dev/null | grep -v amdgpu > gpurun_out/hash_on.log
diff gpurun_out/hash_off.log gpurun_out/hash_on.log > /dev/null && echo "RESULTS IDENTICAL with $SW" || echo "RESULTS DIFFER with $SW"
for rep in 1 2; do
  for e in "X=0" "$SW"; do
    echo "== $e"
    env $e timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --streams 1 2>/dev/null | tail -1 | python tools/bench_kernels.py "$@"
  done
done
