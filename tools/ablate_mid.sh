#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/ablate_mid.log; : > $L
export BQ_MID=1 BQ_MID_PF=2 BQ_MID_ONESHOT=1
for d in 0 32 40 36; do
  echo "=== BQ_DBG=$d" >> $L
  BQ_DBG=$d timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "sepconv_k728_n728_19x19" >> $L
done
cat $L
