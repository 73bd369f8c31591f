#!/bin/bash
# parity of the bf16 path (per layer, end to end) and the per-kernel timing table
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "every_layer or end_to_end or ragged or golden or full_size or full_mode" 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/pytest_wide.log 2>&1
( timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --streams 1 2>gpurun_out/bench_wide.err | tail -1 ) > gpurun_out/bench_wide.json
tail -12 gpurun_out/pytest_wide.log | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/bench_wide.json').read())
    print('value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'streams', d['config']['hip_streams'], 'full', d.get('full_mode_value'))
    print('roofline', d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'], 4), 'frac', round(d['roofline']['frac'], 3))
    for k in d['kernels']:
        print(f"  {k['name']:34s} {k['ms_per_launch']:.4f} ms x{k['launches_per_step']:.0f}  share {k['share']:.3f}  {k['tflops']:.0f} TF  {k['gbps']:.0f} GB/s")
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/bench_wide.err').read()[-1500:])
PY
