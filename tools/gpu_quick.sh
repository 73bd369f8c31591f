#!/bin/bash
# Quick GPU check of a kernel change: layer parity + end-to-end tests, then a short one-stream bench with the per-kernel table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tee gpurun_out/quick_pytest.log | tail -15
timeout 600 python bench.py --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err
tail -c 400 gpurun_out/quick_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/quick_bench.json'))
print('value',d['value'],'ms_per_step',d['ms_per_step'],'roofline',d['roofline']['frac'],d['roofline']['avg_launch_ms'])
for k in d['kernels'][:12]: print(k['name'],round(k['ms_per_launch'],4),k['launches_per_step'])
PY
