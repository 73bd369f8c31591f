#!/bin/bash
for w in 1 2 3; do
  echo "=== BQ_TILE_WGS=$w (mask 15)"
  BQ_TILE_MASK=15 BQ_TILE_WGS=$w timeout 100 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -E "time dtype|conv3x3|sepconv_k64_n128|sepconv_k128_n128|sepconv_k128_n256"
done
