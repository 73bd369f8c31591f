#!/bin/bash
# head split A/B on one box: BQ_HEAD_KS=1024 (32-row tiles) vs default 512 (64-row tiles)
mkdir -p gpurun_out; L=gpurun_out/ab_head.log; : > $L
for rep in 1 2 3; do
  for ks in 4 8; do
    echo "=== WAVES=$ks rep $rep" >> $L
    BQ_HEAD_WAVES=$ks timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|mc_head" >> $L
  done
done
cat $L
