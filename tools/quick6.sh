#!/bin/bash
mkdir -p gpurun_out
echo "--- parity with BQ_SUB=2, n=6"
BQ_SUB=2 timeout 200 python tools/gpu_probe.py e2e --dtype bf16 --n 6 --mc 3 2>&1 | grep -E "e2e|head=="
BQ_SUB=2 timeout 200 python tools/gpu_probe.py e2e --dtype f32 --n 5 --mc 3 2>&1 | grep -E "e2e|head=="
for sbatch in 0 16 32 64; do
  echo "=== BQ_SUB=$sbatch"
  BQ_SUB=$sbatch timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|sepconv_k128_n128_147|sepconv_k64_n128|conv3x3|maxpool_add_147|stem_conv1|sepconv_k256_n256"
done
