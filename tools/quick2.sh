#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/quick.log; : > $L
run() { echo "=== $*" >> $L; timeout 150 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -60 >> $L; echo "rc=${PIPESTATUS[0]}" >> $L; }
run python tools/gpu_probe.py layers --dtype bf16 --n 2
grep -q "features" $L || { tail -20 $L; exit 1; }
run python tools/gpu_probe.py e2e --dtype bf16 --n 3 --mc 4
grep -E "block4_sepconv|block5_out|block12_out|features|e2e|head==|rc=" $L
for d in 0 1 2 4 3; do
  echo "=== BQ_DBG=$d"
  BQ_DBG=$d timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|sepconv_k728_n728_19x19|sepconv_k728_n728_37|sepconv_k256_n728"
done
