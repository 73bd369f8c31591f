"""Read one bench.py JSON line from stdin; print the step time and the kernels whose names contain any of argv[1:]."""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', round(d['ms_per_step'], 3))
pats = sys.argv[1:]
for k in d['kernels']:
    if not pats or any(t in k['name'] for t in pats):
        print(f"  {k['name']:34s} {k['ms_per_launch']:.4f}")
