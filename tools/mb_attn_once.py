#!/usr/bin/env python3
"""A few launches of the window-attention kernels at the README shape (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
qkv = torch.randn(T, 3 * C, device=dev); da = torch.randn(T, C, device=dev); dqkv = torch.empty(T, 3 * C, device=dev)
a = torch.empty(T, C, device=dev)
table = torch.randn(225, heads, device=dev) * 0.02
bT = torch.empty(heads, 64, 64, device=dev); bN = torch.empty(heads, 64, 64, device=dev)
ops.bias_expand(table, bT, bN)
dbT = torch.zeros(heads, 64, 64, device=dev)
for _ in range(4):
    ops.window_attention_fwd(qkv, a, bT, B, H, W, C, heads, 4)
    ops.window_attention_bwd(qkv, da, dqkv, bT, bN, dbT, B, H, W, C, heads, 4)
torch.cuda.synchronize()
