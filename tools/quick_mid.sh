#!/bin/bash
# mid-kernel checks: parity (layers + e2e) in one-shot / persistent / multi-tile-per-WG forms, then timing
mkdir -p gpurun_out; L=gpurun_out/quick4.log; : > $L
export BQ_MID=1 BQ_MID_PF=${PF:-2}
run() { echo "=== $*" >> $L; timeout 200 "$@" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -60 >> $L; echo "rc=${PIPESTATUS[0]}" >> $L; }
BQ_MID_ONESHOT=1 run python tools/gpu_probe.py layers --dtype bf16 --n 2
grep -q "features" $L || { tail -20 $L; exit 1; }
BQ_MID_ONESHOT=1 run python tools/gpu_probe.py e2e --dtype bf16 --n 5 --mc 4
BQ_MID_WGS=3 run python tools/gpu_probe.py e2e --dtype bf16 --n 5 --mc 4
BQ_MID_ONESHOT=1 run python tools/gpu_probe.py time --dtype bf16 --n 256
run python tools/gpu_probe.py time --dtype bf16 --n 256
grep -E "===|features|block[5-9]_out|block1[0-3]_|e2e|head==|time dtype|rc=|sepconv_k728_n728_19" $L
