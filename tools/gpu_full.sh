#!/bin/bash
# all GPU tests, smoke, default bench
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -150 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 ) > gpurun_out/smoke.log
( time timeout 900 python bench.py 2>gpurun_out/bench.err | tail -1 ) > gpurun_out/bench.json 2>gpurun_out/bench.time
grep -n "passed\|failed\|FAILED\|hard weights\|bf16 HIP\|real" gpurun_out/pytest_gpu.log | cut -c1-300; cat gpurun_out/smoke.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'full_mode_value', 'with_reinhard_value', 'f32_value')})
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms', 'traffic')})
print('b1', d.get('b1_latency')); print('tfrecords', d.get('tfrecords')); print('cpu', d.get('cpu_baseline'))
for k in d['kernels']: print(f"  {k['name']:34s} {k['ms_per_launch']:.4f} ms x{k['launches_per_step']:.0f}  share {k['share']:.3f}")
PY
cat gpurun_out/bench.time
