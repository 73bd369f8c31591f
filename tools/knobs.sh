#!/bin/bash
# env-knob sweep of the default bench (one line per setting)
run() { python bench.py --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(round(b['value']), round(b['ms_per_step'],2), b['config']['hip_streams'])"; }
echo "default: $(run)"
for kv in BQ_TILE_MASK=15 BQ_TILE_WGS=1 BQ_NO_SPLIT=1 BQ_NO_TILE=1 BQ_MID=1 BQ_NO_PIPE=1 BQ_SPLIT=1; do
  echo "$kv: $(env $kv bash -c "$(declare -f run); run")"
done
echo "default again: $(run)"
