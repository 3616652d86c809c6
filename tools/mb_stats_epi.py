#!/usr/bin/env python3
"""What the fused epilogue pieces of the proj / fc2 forward GEMM cost: plain, + residual, + row statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T, dev = 32768, "cuda"
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (N, K) in ((180, 180), (180, 360)):
    A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    R = torch.randn(T, N, device=dev); out = torch.empty(T, N, device=dev); st = torch.empty(T, 2, device=dev)
    rs = torch.ones(8, device=dev)
    Wb = ops.split_bf16x3(W)
    for rnd in range(2):
        t0 = timeit(lambda: ops.gemm_nt(A, Wb, b, out=out, epi=0))
        t1 = timeit(lambda: ops.gemm_nt(A, Wb, b, out=out, epi=2, R=R, rowscale=rs, rows_per_scale=4096))
        t2 = timeit(lambda: ops.gemm_nt(A, Wb, b, out=out, epi=2, R=R, rowscale=rs, rows_per_scale=4096, stats_out=st))
        print(f"N={N} K={K}: plain {t0:6.1f}  +residual {t1:6.1f}  +residual+stats {t2:6.1f} us")
