#!/usr/bin/env python3
"""The fused W-MSA forward (wmsa_f16.hip) against the three launches it replaces, README SwinIR block shape at
B = 8: 32768 tokens of 64x64 patches, C = 180, 6 heads.  Rotating buffers (cold operands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
x = torch.randn(T, C, device=dev)
wq = torch.randn(3 * C, C, device=dev) * 0.1; wp = torch.randn(C, C, device=dev) * 0.1
bq = torch.randn(3 * C, device=dev); bp = torch.randn(C, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
table = torch.randn(225, heads, device=dev) * 0.5
Pq, Pp = ops.Bx3(3 * C, C, dev), ops.Bx3(C, C, dev)
bqf = torch.empty(3 * C, device=dev)
tb = ops.PrepTable()
tb.linear(wq, Pq, gamma=gamma); tb.linear(wp, Pp); tb.fold_bias(wq, bq, beta, bqf)
tb.build(dev).run()
biasF, biasG = torch.empty(heads, 64, 64, device=dev), torch.empty(heads, 64, 64, device=dev)
ops.bias_expand_f16(table, biasF, biasG)
st = torch.empty(T, 2, device=dev); ops.layernorm_fwd(x, st)
nb = 6
xs = [x.clone() for _ in range(nb)]
qkvs = [torch.empty(T, 3 * C, device=dev) for _ in range(nb)]
atts = [torch.empty(T, C, device=dev) for _ in range(nb)]
outs = [torch.empty(T, C, device=dev) for _ in range(nb)]
sts = [torch.empty(T, 2, device=dev) for _ in range(nb)]

def timeit(f, n=60):
    for i in range(6): f(i % nb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i % nb)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def fused(shift):
    def f(i): ops.wmsa_fwd_f16(xs[i], st, Pq, bqf, Pp, bp, biasF, qkvs[i], atts[i], outs[i], B, H, W, heads, shift, stats_out=sts[i])
    return f
def sep(shift):
    def f(i):
        ops.gemm_nt(xs[i], Pq, bqf, out=qkvs[i], a_mode=1, ln_stats=st)
        ops.window_attention_fwd_f16(qkvs[i], atts[i], biasF, B, H, W, C, heads, shift)
        ops.gemm_nt(atts[i], Pp, bp, out=outs[i], epi=2, R=xs[i], stats_out=sts[i])
    return f
def qkv_only(i): ops.gemm_nt(xs[i], Pq, bqf, out=qkvs[i], a_mode=1, ln_stats=st)
def att_only(i): ops.window_attention_fwd_f16(qkvs[i], atts[i], biasF, B, H, W, C, heads, 4)
def proj_only(i): ops.gemm_nt(atts[i], Pp, bp, out=outs[i], epi=2, R=xs[i], stats_out=sts[i])

for name, f in [("separate s0", sep(0)), ("separate s4", sep(4)), ("fused s0", fused(0)), ("fused s4", fused(4)),
                ("qkv gemm", qkv_only), ("attention s4", att_only), ("proj gemm", proj_only)] * 2:
    print(f"{name:14s} {timeit(f):7.1f} us", flush=True)
