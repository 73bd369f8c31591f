#!/bin/bash
# round 4: layer parity + e2e on the streaming kernel, then the one-stream bench with the per-kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or end_to_end or ragged or persistent" 2>&1 | tee gpurun_out/stream_pytest.log | tail -15
for rep in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > gpurun_out/stream_bench.json 2> gpurun_out/stream_bench.err
tail -c 300 gpurun_out/stream_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/stream_bench.json'))
print('value',d['value'],'ms_per_step',d['ms_per_step'])
for k in d['kernels'][:14]: print(k['name'],round(k['ms_per_launch'],4),k['launches_per_step'], round(k['gbps']))
PY
done
