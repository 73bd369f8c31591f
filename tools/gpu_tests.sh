#!/bin/bash
# all GPU tests (no -x), log to gpurun_out/pytest_gpu.log
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q -s "$@" 2>&1 | grep -v amdgpu.ids | tail -120 ) > gpurun_out/pytest_gpu.log 2>&1
grep -n "passed\|failed\|FAILED\|hard weights\|bf16 HIP\|real" gpurun_out/pytest_gpu.log | cut -c1-300
