#!/bin/bash
# One GPU-box session: parity tests, smoke, A/B timing, bench, rocprof.  Logs -> gpurun_out/
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/smoke.log
( BQ_NO_PIPE=1 timeout 300 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -v amdgpu.ids | head -14 ) > gpurun_out/time_nopipe.log
( timeout 300 python tools/gpu_probe.py time --dtype bf16 --n 256 2>&1 | grep -v amdgpu.ids | head -14 ) > gpurun_out/time_pipe.log
( timeout 900 python bench.py 2>gpurun_out/bench.err | tail -1 ) > gpurun_out/bench.json
if [ "$1" == "prof" ]; then bash tools/profile.sh ${2:-r01} > gpurun_out/profile.log 2>&1; fi
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; head -4 gpurun_out/time_nopipe.log; head -4 gpurun_out/time_pipe.log; cut -c1-600 gpurun_out/bench.json
