#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, optional rocprof.  Logs -> gpurun_out/
mkdir -p gpurun_out
python - > gpurun_out/cpuinfo.log 2>&1 <<'PY'
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, 'n/a')
PY
( timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30 ) > gpurun_out/pytest_gpu.log
grep -q "passed" gpurun_out/pytest_gpu.log && ! grep -q "failed" gpurun_out/pytest_gpu.log || { cat gpurun_out/pytest_gpu.log | cut -c1-300; echo "PYTEST FAILED - stopping"; exit 1; }
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 ) > gpurun_out/smoke.log
( timeout 600 python bench.py 2>gpurun_out/bench.err | tail -1 ) > gpurun_out/bench.json
if [ "$1" == "prof" ]; then bash tools/profile.sh ${2:-r01} > gpurun_out/profile.log 2>&1; fi
cat gpurun_out/cpuinfo.log; tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; cut -c1-1200 gpurun_out/bench.json; tail -2 gpurun_out/bench.err | cut -c1-300
