#!/bin/bash
# Round 5, GPU session A: (1) the whole -m gpu suite on the new tree; (2) same-box A/B of the LDS layouts -- biscuit_amd/libA.so
# (round 4's A-tile strides) against libB.so (slots per row = 2 mod 4), alternated; (3) the schedules: free-running streams with
# and without mask-sized grids, antiphase, pipeline at several splits; (4) SQ LDS counters of the new layout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r5a; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('$1', 'FAILED', e); sys.exit(0)
ks={k['name']:k['ms_per_launch'] for k in d.get('kernels',[])}
sel=['sepconv_k728_n728_19x19','blocktail_147_c128','blocktail_74_c256','front_stage_stem_conv2','sepconv_k64_n128_147x147','sepconv_k128_n256_74x74']
print('$1', 'step %.3f ms  %.0f tiles/s' % (d['ms_per_step'], d['value']), d['config'].get('schedule'), d['config'].get('cus'), 'in flight', d['config']['hip_streams'],
      ' '.join('%s=%.4f' % (n.split('_',1)[1][:14], ks[n]) for n in sel if n in ks), 'entry_side %.3f' % d.get('entry_side_ms', -1))
"; }
for rep in 1 2 3; do
  for v in A B; do
    cp biscuit_amd/lib$v.so biscuit_amd/libbiscuit_hip.so
    timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 100 2>$O/err_$v.log | show "lds_$v rep$rep" >> $O/ab.log
  done
done
cp biscuit_amd/libB.so biscuit_amd/libbiscuit_hip.so
for rep in 1 2; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 100 --streams 2 2>>$O/err_s.log | show "free2          rep$rep" >> $O/sched.log
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 100 --streams 2 --size-grids 2>>$O/err_s.log | show "free2+sized    rep$rep" >> $O/sched.log
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 100 --streams 4 --size-grids 2>>$O/err_s.log | show "free4+sized    rep$rep" >> $O/sched.log
  timeout 300 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 100 --schedule antiphase 2>>$O/err_s.log | show "antiphase      rep$rep" >> $O/sched.log
  for ce in 96 104 112 120 128; do
    timeout 300 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 100 --schedule pipeline --cus-entry $ce 2>>$O/err_s.log | show "pipeline $ce   rep$rep" >> $O/sched.log
  done
done
cat $O/pytest.log $O/ab.log $O/sched.log
bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1
tail -30 $O/pmc_sq.log
