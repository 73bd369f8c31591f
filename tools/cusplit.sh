#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(round(b['value']), round(b['ms_per_step'],2), b['config']['hip_streams'])"; }
for m in contig xcd interleave; do for i in 1 2; do echo "split=$m run $i: $(BQ_CU_SPLIT=$m run)"; done; done
echo "nosplit: $(run)"
