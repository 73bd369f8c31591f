#!/bin/bash
# error side of an A/B of library builds (ab/<name>.so): tools/parity_seeds.py (8 stress draws x 256 tiles, f16 against the fp32
# kernels) and the config-2-size oracle test per build: bash tools/gpu_ab_err.sh TAG "H0 H3" -> gpurun_out/TAG_<name>_err.txt
T=${1:-ab}; V=${2:-"A B"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp biscuit_amd/libbiscuit_hip.so /tmp/keep.so
for v in $V; do
  cp ab/$v.so biscuit_amd/libbiscuit_hip.so
  (python tools/parity_seeds.py 2>&1 | grep -E "f16|worst" ; python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "config2_size" 2>&1 | grep -E "vs|passed|failed") > gpurun_out/${T}_${v}_err.txt
  echo "== $v"; cat gpurun_out/${T}_${v}_err.txt
done
cp /tmp/keep.so biscuit_amd/libbiscuit_hip.so
