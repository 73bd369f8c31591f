#!/bin/bash
# per-kernel averages of one short one-stream bench run (rocprofv3 --kernel-trace --stats): bash tools/gpu_kstats.sh [pattern]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --no-profile --no-extras > $OUT/log.txt 2>&1
python3 - "$OUT" "${1:-.}" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']):
        print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
