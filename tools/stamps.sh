#!/bin/bash
mkdir -p gpurun_out
BQ_MID=1 BQ_MID_PF=${PF:-2} BQ_MID_ONESHOT=${ONESHOT-1} BQ_STAMPS=gpurun_out/stamps.bin timeout 200 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|k728_n728_19"
python tools/stamps.py gpurun_out/stamps.bin > gpurun_out/stamps.txt 2>&1; tail -12 gpurun_out/stamps.txt
