#!/usr/bin/env python3
"""Slot stamps (100 MHz wall clock) of the producer / consumer MLP forward k_mlp_gr_fwd (mlp_f16.hip): 12 waves per block
(0..7 matrix waves, 8..11 row waves), one block per CU, two tiles per block.  Experiments build (SRHIP_LIB)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

M, C, hid = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 180, 360
dev = "cuda"
x = torch.randn(M, C, device=dev)
w1 = torch.randn(hid, C, device=dev) * 0.1; w2 = torch.randn(C, hid, device=dev) * 0.1
b1 = torch.randn(hid, device=dev); b2 = torch.randn(C, device=dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
P = {k: ops.Bx3(*s, dev) for k, s in dict(w1=(hid, C), w2=(C, hid)).items()}
b1f = torch.empty(hid, device=dev)
tb = ops.PrepTable()
tb.linear(w1, P["w1"], gamma=gamma); tb.linear(w2, P["w2"]); tb.fold_bias(w1, b1, beta, b1f)
tb.build(dev).run()
st = torch.empty(M, 2, device=dev); ops.layernorm_fwd(x, st)
h = torch.empty(M, hid, device=dev); out = torch.empty(M, C, device=dev); sto = torch.empty(M, 2, device=dev)
evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
nblk = (M // 64 + 1) // 2
dbg = torch.zeros(nblk, 12, 32, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_mlp_debug_buffer
fn.argtypes = [ctypes.c_void_p]
def fwd(): ops.mlp_fwd_f16(x, st, P["w1"], b1f, P["w2"], b2, out, h=h, stats_out=sto)
for _ in range(3): fwd()
t = []
for _ in range(9):
    evict.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fwd(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1) * 1e3)
print(f"forward, cold operands: launch median {sorted(t)[4]:6.1f} us, min {min(t):6.1f}")
evict.fill_(1)
fn(dbg.data_ptr()); fwd(); torch.cuda.synchronize(); fn(None)
d = dbg.cpu().double() * 0.01
t0 = d[:, :, 0].min()
G = {0: "start", 1: "W1 requested", 2: "barrier", 3: "gemm1+act t0", 4: "barrier", 5: "a2 t0", 6: "barrier", 7: "gemm2 t0", 8: "barrier",
     9: "tile out t0", 10: "barrier", 11: "gemm1+act t1", 12: "barrier", 13: "a2 t1", 14: "barrier", 15: "gemm2 t1", 16: "barrier",
     17: "tile out t1"}
R = {0: "start", 1: "x t0 staged", 2: "barrier", 3: "x t1 staged", 10: "4 barriers", 11: "epilogue t0", 18: "4 barriers", 19: "epilogue t1"}
for name, ws, names in (("matrix waves", slice(0, 8), G), ("row waves", slice(8, 12), R)):
    print(name)
    prev = None
    for k in sorted(names):
        v = d[:, ws, k] - t0
        print(f"  {k:2d} {names[k]:16s} mean {v.mean():7.2f}  max {v.max():7.2f}" + ("" if prev is None else f"   step {(d[:, ws, k] - d[:, ws, prev]).mean():6.2f}"))
        prev = k
