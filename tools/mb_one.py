#!/usr/bin/env python3
"""Run ONE op shape a few times (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
T, N, K = 32768, int(sys.argv[1]), int(sys.argv[2])
epi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = "cuda"
A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
R = torch.randn(T, N, device=dev); out = torch.empty(T, N, device=dev)
for _ in range(5):
    ops.gemm_nt(A, W, b, out=out, epi=epi, R=R if epi >= 2 else None)
torch.cuda.synchronize()
