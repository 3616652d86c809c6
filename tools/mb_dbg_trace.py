#!/usr/bin/env python3
"""Run under rocprofv3 --kernel-trace: ablation variants of the bf16x3 NT kernel, 5 launches each,
in a fixed order (tools/parse_dbg_trace.py maps trace rows back to variants)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T, dev = 32768, "cuda"
shapes = [(180, 180, 0, 2), (360, 180, 1, 0), (180, 360, 2, 2), (540, 180, 1, 0), (360, 180, 0, 3), (180, 540, 0, 0)]
variants = [0, 8, 4, 12, 16, 32, 48, 56, 58, 62, 2]
data = []
for (N, K, a_mode, epi) in shapes:
    A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    R = torch.randn(T, N, device=dev); out = torch.empty(T, N, device=dev)
    st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    data.append((A, ops.split_bf16x3(W), b, dict(out=out, a_mode=a_mode, ln_stats=st if a_mode == 1 else None, epi=epi,
                                                  R=R if epi >= 2 else None)))
torch.cuda.synchronize()
for dbg in variants:
    os.environ["SRHIP_NT_DBG"] = str(dbg)
    for (A, Wb, b, kw) in data:
        for _ in range(5):
            ops.gemm_nt(A, Wb, b, **kw)
        torch.cuda.synchronize()
print("variants", variants, "shapes", shapes)
