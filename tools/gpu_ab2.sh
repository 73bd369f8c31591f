#!/bin/bash
# Same-box A/B/A/B of two builds of the library (ab/A.so, ab/B.so: git-ignored, they travel with gpurun): bash tools/gpu_ab2.sh TAG
# -> gpurun_out/TAG_{A,B}_kern.txt (two bench runs each per pass, per-kernel tables: tools/gpu_kern.sh)
T=${1:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp biscuit_amd/libbiscuit_hip.so /tmp/keep.so
for rep in 1 2; do
  for v in A B; do
    cp ab/$v.so biscuit_amd/libbiscuit_hip.so
    bash tools/gpu_kern.sh ${T}_${v}${rep} 1 > /dev/null
  done
done
cp /tmp/keep.so biscuit_amd/libbiscuit_hip.so
for v in A B; do for rep in 1 2; do echo "== $v$rep"; head -12 gpurun_out/${T}_${v}${rep}_kern.txt; done; done
