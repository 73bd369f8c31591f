#!/bin/bash
# A/B of two builds of the library on one box: biscuit_amd/libA.so vs biscuit_amd/libB.so (made by hand:
# build, copy libbiscuit_hip.so to libA.so, change, rebuild, copy to libB.so)
mkdir -p gpurun_out; L=gpurun_out/ab.log; : > $L
for rep in 1 2 3; do
  for v in A B; do
    cp biscuit_amd/lib$v.so biscuit_amd/libbiscuit_hip.so
    echo "=== $v rep $rep" >> $L
    timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|147x147" >> $L
  done
done
cat $L
