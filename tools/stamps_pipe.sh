#!/bin/bash
mkdir -p gpurun_out
BQ_STAMPS_PIPE=gpurun_out/stamps_pipe_res.bin timeout 200 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "k728_n728_19"
python tools/stamps.py gpurun_out/stamps_pipe_res.bin 2>/dev/null | grep -E "avg|half" 
BQ_STAMPS_NORES=1 BQ_STAMPS_PIPE=gpurun_out/stamps_pipe_nores.bin timeout 200 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "k728_n728_19"
python tools/stamps.py gpurun_out/stamps_pipe_nores.bin 2>/dev/null | grep -E "avg|half"
