#!/bin/bash
# Same-box A/B/A/B of several builds of the library (ab/<name>.so: git-ignored, they travel with gpurun):
#   bash tools/gpu_ab2.sh TAG "A B ..." [reps] [lines]   -> gpurun_out/TAG_<name><rep>_kern.txt (per-kernel tables: tools/gpu_kern.sh)
T=${1:-ab}; V=${2:-"A B"}; REPS=${3:-2}; LINES=${4:-12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp biscuit_amd/libbiscuit_hip.so /tmp/keep.so
for rep in $(seq 1 $REPS); do
  for v in $V; do
    cp ab/$v.so biscuit_amd/libbiscuit_hip.so
    bash tools/gpu_kern.sh ${T}_${v}${rep} 1 > /dev/null
  done
done
cp /tmp/keep.so biscuit_amd/libbiscuit_hip.so
for v in $V; do for rep in $(seq 1 $REPS); do echo "== $v$rep"; head -$LINES gpurun_out/${T}_${v}${rep}_kern.txt; done; done
