#!/usr/bin/env python3
"""Fused MLP kernels on the two-plane fp16 operands (mlp_f16.hip) against the Linear launches they replace, README
SwinIR block shape at B = 8: M = 32768 tokens, 180 -> 360 -> 180.  Rotating buffers (cold operands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

M, C, hid = 32768, 180, 360
dev = "cuda"
x = torch.randn(M, C, device=dev); dy = torch.randn(M, C, device=dev)
w1 = torch.randn(hid, C, device=dev) * 0.1; w2 = torch.randn(C, hid, device=dev) * 0.1
b1 = torch.randn(hid, device=dev); b2 = torch.randn(C, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
P = {k: ops.Bx3(*s, dev) for k, s in dict(w1=(hid, C), w2=(C, hid), w1T=(C, hid), w2T=(hid, C)).items()}
b1f = torch.empty(hid, device=dev)
tb = ops.PrepTable()
tb.linear(w1, P["w1"], gamma=gamma); tb.linear(w1, P["w1T"], gamma=gamma, transpose=True)
tb.linear(w2, P["w2"]); tb.linear(w2, P["w2T"], transpose=True)
tb.fold_bias(w1, b1, beta, b1f)
tb.build(dev).run()
st = torch.empty(M, 2, device=dev); ops.layernorm_fwd(x, st)
nb = 6
xs = [x.clone() for _ in range(nb)]; hs = [torch.empty(M, hid, device=dev) for _ in range(nb)]
outs = [torch.empty(M, C, device=dev) for _ in range(nb)]; sts = [torch.empty(M, 2, device=dev) for _ in range(nb)]
dhs = [torch.empty(M, hid, device=dev) for _ in range(nb)]; ghs = [torch.empty(M, hid, device=dev) for _ in range(nb)]

def timeit(f, n=60):
    for i in range(6): f(i % nb)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i % nb)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def fwd_fused(i): ops.mlp_fwd_f16(xs[i], st, P["w1"], b1f, P["w2"], b2, outs[i], h=hs[i], stats_out=sts[i])
def fwd_fused_eval(i): ops.mlp_fwd_f16(xs[i], st, P["w1"], b1f, P["w2"], b2, outs[i], stats_out=sts[i])
def fwd_sep(i):
    ops.gemm_nt(xs[i], P["w1"], b1f, out=hs[i], a_mode=1, ln_stats=st)
    ops.gemm_nt(hs[i], P["w2"], b2, out=outs[i], a_mode=2, epi=2, R=xs[i], stats_out=sts[i])
def bwd_fused(i): ops.mlp_bwd_f16(xs[i], P["w2T"], P["w1T"], hs[i], dhs[i], ghs[i], xs[(i + 1) % nb], st, outs[i])
def bwd_sep(i):
    ops.gemm_nt(xs[i], P["w2T"], None, out=dhs[i], epi=3, R=hs[i], aux=ghs[i])
    ops.gemm_nt_lnbwd(dhs[i], P["w1T"], xs[(i + 1) % nb], st, xs[i], outs[i])

for i in range(nb): fwd_sep(i)
for name, f in [("fwd separate", fwd_sep), ("fwd fused", fwd_fused), ("fwd fused (no h)", fwd_fused_eval),
                ("bwd separate", bwd_sep), ("bwd fused", bwd_fused)] * 2:
    print(f"{name:18s} {timeit(f):7.1f} us", flush=True)
