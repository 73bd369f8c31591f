#!/bin/bash
# experiments build on the box; steady-state stamps of the FIRST wide-kernel launch without a residual (block3_sepconv2)
mkdir -p gpurun_out
make -C biscuit_amd/csrc clean >/dev/null 2>&1
make -C biscuit_amd/csrc -j16 EXPERIMENTS=1 EXPFLAGS=-DWIDE_ABLATE=${2:-0} 2>&1 | grep -E "error|FAILED" | head -3
BQ_STAMPS_NORES=1 BQ_STAMPS_B0=${1:-64} BQ_STAMPS_WIDE=gpurun_out/stamps_b3.bin timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 4 --streams 1 2>gpurun_out/b3_err.log | tail -c 200; tail -5 gpurun_out/b3_err.log
python - <<'PY'
import numpy as np
a = np.fromfile('gpurun_out/stamps_b3.bin', dtype=np.uint64).reshape(64, 8, 8, 32).astype(np.int64)
ok = a[:, :, 0, 0] > 0
print('workgroups x waves with stamps:', int(ok.sum()))
t = a[ok]
med = lambda x: int(np.median(x))
print('prologue: ', [med(t[:, 0, e + 1] - t[:, 0, e]) for e in range(5)])
for it in range(8):
    s = t[:, it]
    if not (s[:, 22] > 0).all(): break
    nch = 4
    chunks = [med(s[:, 6] - s[:, 18])] + [med(s[:, 6 + c] - s[:, 5 + c]) for c in range(1, nch)]
    print(f'tile {it}: total {med(s[:, 22] - s[:, 18])} chunks {chunks} nops {med(s[:, 20] - s[:, 5 + nch])} epilogue {med(s[:, 22] - s[:, 20])}')
PY
