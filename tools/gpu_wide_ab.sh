#!/bin/bash
# A/B of kernels_wide.hip variants on ONE box: the tree's build first, then one rebuild per argument (each a -D flag list).
# AB_FILE=<source to rebuild, default kernels_wide.hip> AB_SHOW="<substrings of layer classes to print>"
# usage: bash tools/gpu_wide_ab.sh "-DWIDE_TOUCH=0" ...      prints per-layer ms of the wide-kernel layers for each build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 60 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ms_per_step %.4f' % d['ms_per_step'], ' '.join('%s=%.4f' % (k['name'].replace('sepconv_',''), k['ms_per_launch']) for k in d['kernels'] if any(t in k['name'] for t in '${AB_SHOW:-n728 k256_n256}'.split())))
"
}
echo "== tree build"; run; run
for f in "$@"; do
  touch biscuit_amd/csrc/${AB_FILE:-kernels_wide.hip}
  SECONDS=0
  make -C biscuit_amd/csrc -j16 EXTRA="$f" > gpurun_out/ab_make.log 2>&1; echo "  rebuild rc=$? ${SECONDS}s: $(tail -1 gpurun_out/ab_make.log | cut -c1-150)"
  echo "== $f"; run; run
done
