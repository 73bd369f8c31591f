#!/bin/bash
# A/B of an environment knob on one box: tools/ab_env.sh VAR=VALUE 'grep pattern'
mkdir -p gpurun_out; L=gpurun_out/ab_env.log; : > $L
for rep in 1 2 3; do
  echo "=== default rep $rep" >> $L
  timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|$2" >> $L
  echo "=== $1 rep $rep" >> $L
  env $1 timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|$2" >> $L
done
cat $L
