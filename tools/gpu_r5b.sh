#!/bin/bash
# Round 5, GPU session B: (1) the -m gpu suite (new: activation exponents); (2) A/B/C of the A-tile strides: libA round 4, libB
# pad 32 everywhere, libC pad 16 for the 128 -> 256 streaming instance only; (3) kernel traces of the schedules for
# tools/pairing.py: what runs next to a 19x19 launch and how long it then takes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r5b; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 ) > $O/pytest.log 2>&1
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
cp biscuit_amd/libbiscuit_hip.so biscuit_amd/libKEEP.so
for rep in 1 2 3; do
  for v in A B C; do
    cp biscuit_amd/lib$v.so biscuit_amd/libbiscuit_hip.so
    timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 100 2>$O/err_$v.log | show "lds_$v rep$rep" >> $O/ab.log
  done
done
cp biscuit_amd/libKEEP.so biscuit_amd/libbiscuit_hip.so
cd /tmp && export TMPDIR=/tmp
for sched in "free1:--streams 1" "free2:--streams 2 --fixed-streams" "antiphase:--schedule antiphase" "pipeline128:--schedule pipeline --cus-entry 128"; do
  tag=${sched%%:*}; flags=${sched#*:}
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-profile --steps 60 $flags > $O/trace_$tag.log 2>&1
  f=$(ls $O/trace_$tag/*/*_kernel_trace.csv 2>/dev/null | head -1)
  tail -1 $O/trace_$tag.log | show "traced_$tag" >> $O/pair.log
  [ -n "$f" ] && python3 $R/tools/pairing.py $f $tag >> $O/pair.log 2>&1
  rm -rf $O/trace_$tag     # (tens of MB per trace: only the table travels back)
done
cd $R
cat $O/pytest.log $O/ab.log $O/pair.log
