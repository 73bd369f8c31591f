#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/ablate2.log; : > $L
for d in 0 128 129 144 145; do
  echo "=== BQ_DBG=$d" >> $L
  BQ_DBG=$d timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|sepconv_k728_n728_19x19|sepconv_k728_n728_37|sepconv_k256_n728" >> $L
done
cat $L
