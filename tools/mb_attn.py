#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops
B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
qkv = torch.randn(T, 3 * C, device=dev); da = torch.randn(T, C, device=dev); dqkv = torch.empty(T, 3 * C, device=dev)
a = torch.empty(T, C, device=dev)
table = torch.randn(225, heads, device=dev) * 0.02
bT = torch.empty(heads, 64, 64, device=dev); bN = torch.empty(heads, 64, 64, device=dev)
ops.bias_expand(table, bT, bN)
dbT = torch.zeros(heads, 64, 64, device=dev)
for shift in (0, 4):
    tf = timeit(lambda: ops.window_attention_fwd(qkv, a, bT, B, H, W, C, heads, shift))
    tb = timeit(lambda: ops.window_attention_bwd(qkv, da, dqkv, bT, bN, dbT, B, H, W, C, heads, shift))
    print(f"shift {shift}: fwd {tf:6.1f} us   bwd (q + kv) {tb:6.1f} us   env NOATOMIC={os.environ.get('SRHIP_WA_NOATOMIC')}")
