"""What the vendor GEMM (hipBLASLt / rocBLAS behind torch.matmul) reaches on this box, as a calibration of the MFMA "peak" the
roofline fractions are priced against: the 728 -> 728 @19x19 layer's pointwise shape (M = 256 x 361, K = N = 728) and two
large square shapes, f16 and bf16, random normal data, fp32 accumulation.  GPU box only; prints one line per case.
usage: python tools/gemm_ref.py [iters]"""
import sys
import time

import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = 'cuda'
cases = [('layer 728->728 @19x19 x256', 256 * 361, 728, 728), ('layer padded 736/768', 256 * 361, 768, 736),
         ('block13 728->1024 @19x19', 256 * 361, 1024, 728), ('block14 1024->1536 @10x10', 25600, 1536, 1024),
         ('block14 1536->2048 @10x10', 25600, 2048, 1536), ('block4 256->728 @37x37', 256 * 1369, 728, 256),
         ('4096^3', 4096, 4096, 4096), ('8192^3', 8192, 8192, 8192)]
for dt in (torch.float16, torch.bfloat16):
    for name, m, n, k in cases:
        a = torch.randn(m, k, device=dev, dtype=dt)
        b = torch.randn(k, n, device=dev, dtype=dt)
        for _ in range(10):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * m * n * k / ms / 1e9
        print(f'{str(dt)[6:]:9s} {name:28s} M={m} N={n} K={k}: {ms:.4f} ms  {tf:7.1f} TFLOP/s = {tf / 2500:.3f} of 2.5 PF', flush=True)
