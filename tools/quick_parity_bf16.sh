#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/quick.log; : > $L
run() { echo "=== $*" >> $L; timeout 150 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -60 >> $L; echo "rc=${PIPESTATUS[0]}" >> $L; }
run python tools/gpu_probe.py layers --dtype bf16 --n 2
grep -q "features" $L || { tail -20 $L; exit 1; }
run python tools/gpu_probe.py e2e --dtype bf16 --n 3 --mc 4
run python tools/gpu_probe.py time --dtype bf16 --n 256
head -16 $L; grep -E "features|e2e|head==|time dtype|rc=" $L; grep -A 14 "time dtype" $L | tail -14
