#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T = 32768
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(180, 180, 0, 2), (360, 180, 1, 0), (180, 360, 2, 2), (540, 180, 1, 0), (360, 180, 0, 3), (180, 540, 0, 0)]
for stg in [0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16]:
    os.environ["SRHIP_NTB_STAGGER"] = str(stg)
    row = []
    for (N, K, a_mode, epi) in shapes:
        A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
        R = torch.randn(T, N, device=dev); out = torch.empty(T, N, device=dev)
        st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        Wb = ops.split_bf16x3(W)
        kw = dict(out=out, a_mode=a_mode, ln_stats=st if a_mode == 1 else None, epi=epi, R=R if epi >= 2 else None)
        row.append(timeit(lambda: ops.gemm_nt(A, Wb, b, **kw)))
    print(f"stagger={stg:3d} (x512 cyc): " + "  ".join(f"{t:6.1f}" for t in row) + f"   sum {sum(row):6.1f}")
