#!/bin/bash
# round 2, first GPU session: all GPU tests (no -x: see every failure), bf16 error on the stress weights, bench
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -60 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 600 python tools/bf16_error.py 128 2>&1 | grep -v amdgpu.ids ) > gpurun_out/bf16_error.log
( time timeout 900 python bench.py 2>gpurun_out/bench.err | tail -1 ) > gpurun_out/bench.json 2>gpurun_out/bench.time
tail -40 gpurun_out/pytest_gpu.log | cut -c1-400; cat gpurun_out/bf16_error.log; cut -c1-3000 gpurun_out/bench.json; tail -3 gpurun_out/bench.err | cut -c1-300; cat gpurun_out/bench.time
