#!/bin/bash
mkdir -p gpurun_out; L=gpurun_out/ablate_mid.log; : > $L
BQ_MID_PF=2 BQ_MID_ONESHOT=1 timeout 100 python tools/gpu_probe.py e2e --dtype bf16 --n 5 --mc 4 >> $L 2>&1
BQ_MID_PF=2 BQ_MID_WGS=3 timeout 100 python tools/gpu_probe.py e2e --dtype bf16 --n 5 --mc 4 >> $L 2>&1
for pf in 2 4; do
for d in 0 4; do
  echo "=== BQ_DBG=$d oneshot_pf$pf" >> $L
  BQ_MID_PF=$pf BQ_MID_ONESHOT=1 BQ_DBG=$d timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "sepconv_k728_n728_19x19" >> $L
  echo "=== BQ_DBG=$d persist_pf$pf" >> $L
  BQ_MID_PF=$pf BQ_DBG=$d timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|sepconv_k728_n728_19x19" >> $L
done
done
cat $L
