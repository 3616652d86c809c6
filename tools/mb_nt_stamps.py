#!/usr/bin/env python3
"""s_memtime stamps of the bf16x3 NT GEMM K loop (SRHIP_NT_DBG=64): cycles per phase of wave 0 of two blocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T, dev = 32768, "cuda"
os.environ["SRHIP_NT_DBG"] = "64"
for (N, K, a_mode, epi) in [(180, 180, 0, 2), (360, 180, 1, 0), (180, 360, 2, 2), (540, 180, 1, 0), (180, 540, 0, 0)]:
    A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    R = torch.randn(T, N, device=dev); out = torch.zeros(T, N, device=dev)
    st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    Wb = ops.split_bf16x3(W)
    for _ in range(3):
        ops.gemm_nt(A, Wb, b, out=out, a_mode=a_mode, ln_stats=st if a_mode == 1 else None, epi=epi, R=R if epi >= 2 else None)
    torch.cuda.synchronize()
    o = out.flatten()[:16].cpu().tolist()
    for blk, v in (("first", o[:8]), ("mid", o[8:])):
        n = max(v[6], 1)
        print(f"N={N:3d} K={K:3d} a={a_mode} {blk:5s} block: prologue {v[0]:6.0f} | per chunk: barrier1 {v[1]/n:6.0f} stage {v[2]/n:6.0f} "
              f"barrier2 {v[3]/n:6.0f} load-issue {v[4]/n:6.0f} mfma {v[5]/n:6.0f}  (chunks {v[6]:.0f}; 36 MFMA = 1152 cycles)")
