cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg3 -- python3 $R/bench.py --workload cfg3 --slides 24 --tiles-per-slide 1000 --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/prof_cfg3.log 2>&1
tail -1 $R/gpurun_out/prof_cfg3.log | cut -c1-200
python3 - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=sorted(glob.glob(R+'/gpurun_out/prof_cfg3/*/*_kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in rows)
n=len(ev); seg=ev[int(n*0.4):int(n*0.95)]
span=seg[-1][1]-seg[0][0]
# union of intervals
cur_s,cur_e=seg[0][0],seg[0][1]; busy=0
for s,e,_ in seg[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print('span ms', span/1e6, 'union busy ms', busy/1e6, 'idle %', 100*(1-busy/span))
from collections import Counter
c=Counter(); t=Counter()
for s,e,k in seg: c[k]+=1; t[k]+=e-s
for k,v in t.most_common(8): print(f'{v/1e6:9.2f} ms {c[k]:6d} {k}')
PY
