#!/bin/bash
mkdir -p gpurun_out
for d in 0 1 2 16 18; do
  BQ_DBG=$d BQ_STAMPS_NORES=1 BQ_STAMPS_PIPE=gpurun_out/stamps_pipe_dbg$d.bin timeout 200 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "k728_n728_19" | awk '{print "dbg='$d'", $4, $5}'
  python tools/stamps.py gpurun_out/stamps_pipe_dbg$d.bin 2>/dev/null | grep -E "per chunk|half"
done
