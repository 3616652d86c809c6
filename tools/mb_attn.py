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
bF = torch.empty(heads, 64, 64, device=dev); bG = torch.empty(heads, 64, 64, device=dev)
ops.bias_expand_f16(table, bF, bG)
# rotating operand sets: the step never finds qkv in L2 / the Infinity Cache
qs = [torch.randn(T, 3 * C, device=dev) for _ in range(6)]
for shift in (0, 4):
    it = [0]
    def f32():
        it[0] += 1; ops.window_attention_fwd(qs[it[0] % 6], a, bT, B, H, W, C, heads, shift)
    def f16():
        it[0] += 1; ops.window_attention_fwd_f16(qs[it[0] % 6], a, bF, B, H, W, C, heads, shift)
    def b32():
        it[0] += 1; ops.window_attention_bwd(qs[it[0] % 6], da, dqkv, bT, bN, dbT, B, H, W, C, heads, shift)
    def b16():
        it[0] += 1; ops.window_attention_bwd_f16(qs[it[0] % 6], da, dqkv, bF, bG, dbT, B, H, W, C, heads, shift)
    print(f"shift {shift}: fwd exact-f32 MFMA {timeit(f32):6.1f} us   fwd fp16x2 {timeit(f16):6.1f} us   "
          f"bwd exact-f32 {timeit(b32):6.1f} us   bwd fp16x2 {timeit(b16):6.1f} us (cold operands)")
for shift in (0, 4):
    tf = timeit(lambda: ops.window_attention_fwd(qkv, a, bT, B, H, W, C, heads, shift))
    tb = timeit(lambda: ops.window_attention_bwd(qkv, da, dqkv, bT, bN, dbT, B, H, W, C, heads, shift))
    print(f"shift {shift}: fwd {tf:6.1f} us   bwd (q + kv) {tb:6.1f} us   env NOATOMIC={os.environ.get('SRHIP_WA_NOATOMIC')}")
