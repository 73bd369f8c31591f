#!/bin/bash
# board power / clocks / temperature while the one-stream bench runs (rocm-smi samples every 2 s), then idle
mkdir -p gpurun_out
timeout 600 python bench.py --steps 1500 --warmup 20 --streams 1 --no-cpu-baseline --no-extras > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge|memory|hbm)" | tr '\n' ';' | cut -c1-600
  echo
  sleep 2
done
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
wait $BP
python - <<'PY'
import json
d=json.load(open('gpurun_out/power_bench.json'))
print('value', d['value'], 'ms_per_step', d['ms_per_step'])
PY
sleep 5
echo idle:
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ';'
