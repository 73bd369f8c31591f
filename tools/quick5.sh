#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/quick.log; : > $L
run() { echo "=== $*" >> $L; timeout 150 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -60 >> $L; echo "rc=${PIPESTATUS[0]}" >> $L; }
run python tools/gpu_probe.py layers --dtype bf16 --n 2
grep -q "features" $L || { tail -20 $L; exit 1; }
run python tools/gpu_probe.py time --dtype bf16 --n 256
grep -E "block1_conv2|block2_sepconv|features|time dtype|rc=" $L; grep -A 12 "time dtype" $L | tail -12
rm -rf /tmp/cli_out; timeout 200 python -m biscuit_amd --synthetic 4x8 --mc 5 --out /tmp/cli_out 2>&1 | grep -v amdgpu | tail -2; ls /tmp/cli_out
