#!/bin/bash
# run every probe stage in its own process so one fault does not hide the rest
mkdir -p gpurun_out
L=gpurun_out/probe.log
: > $L
run() { echo "=== $*" >> $L; timeout 300 python tools/gpu_probe.py "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
rocminfo | grep -E "Marketing|gfx" | head -4 >> $L 2>&1
run stage --dtype f32 --n 3
run stage --dtype bf16 --n 3
run head --dtype f32 --n 5 --mc 7
run layers --dtype f32 --n 2
run layers --dtype bf16 --n 2
run e2e --dtype f32 --n 3 --mc 4
run e2e --dtype bf16 --n 3 --mc 4
run time --dtype bf16 --n 64
run time --dtype bf16 --n 256
run time --dtype f32 --n 32
tail -c 3000 $L
