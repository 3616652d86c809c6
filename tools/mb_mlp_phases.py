#!/usr/bin/env python3
"""Wall-clock stamps (100 MHz) of the phases of the fused MLP kernels (mlp_f16.hip), all waves of all blocks.  Needs an
experiment build (make -C sr-caco-2_amd/csrc EXPERIMENTS=1 OUT=../lib/libsrhip_exp.so OBJDIR=../lib/obj_exp) loaded
through SRHIP_LIB."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

M, C, hid = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 180, 360
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
h = torch.empty(M, hid, device=dev); out = torch.empty(M, C, device=dev); sto = torch.empty(M, 2, device=dev)
dh = torch.empty(M, hid, device=dev); gh = torch.empty(M, hid, device=dev); dx = torch.empty(M, C, device=dev)
nblk = M // 64
dbg = torch.zeros(nblk, 4, 32, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_mlp_debug_buffer
fn.argtypes = [ctypes.c_void_p]
# the chained form of the training step: front = the qkv Linear's data gradient (K0 = 3 C) + LayerNorm backward of the block
# behind, chain = the proj Linear's data gradient of this block
K0 = 3 * C
X0 = torch.randn(M, K0, device=dev); x0 = torch.randn(M, C, device=dev); res0 = torch.randn(M, C, device=dev)
st0 = torch.empty(M, 2, device=dev); ops.layernorm_fwd(x0, st0)
w0 = torch.randn(K0, C, device=dev) * 0.1; w3 = torch.randn(C, C, device=dev) * 0.1
P["w0T"] = ops.Bx3(C, K0, dev); P["w3T"] = ops.Bx3(C, C, dev)
tb = ops.PrepTable()
tb.linear(w0, P["w0T"], transpose=True); tb.linear(w3, P["w3T"], transpose=True)
tb.build(dev).run()
out3 = torch.empty(M, C, device=dev)

def fwd(): ops.mlp_fwd_f16(x, st, P["w1"], b1f, P["w2"], b2, out, h=h, stats_out=sto)
def bwd(): ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, x, st, dx)
def bwd_full(): ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, x, st, dx, chain=(P["w3T"], out3, None),
                                front=(X0, P["w0T"], x0, st0, res0))
# slot -> name, in time order per direction
ORDER = {"forward": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14],
         "backward": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 14],
         "backward qkv+ +proj": [0, 22, 23, 15, 16, 17, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 19, 20, 21, 14]}
NAMES = {0: "start", 1: "x staged", 2: "barrier", 3: "gemm1 done", 4: "activation done", 5: "max+barrier", 6: "h pass 1 staged",
         7: "barrier", 8: "gemm2 pass 1", 9: "barrier", 10: "h pass 2 staged+barrier", 11: "gemm2 pass 2", 12: "barrier",
         13: "T laid+barrier", 14: "end", 15: "front gemm (3 passes)", 16: "front T laid+barriers", 17: "front LN bwd rows",
         18: "LN bwd rows", 22: "front pass 0 rows staged", 23: "front pass 0 products", 19: "dx images+barriers", 20: "gemm3", 21: "T laid+barriers"}
for name, f in (("forward", fwd), ("backward", bwd), ("backward qkv+ +proj", bwd_full)):
    for _ in range(3):
        f()
    t = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1) * 1e3)
    fn(dbg.data_ptr()); f(); torch.cuda.synchronize(); fn(None)
    d = dbg.cpu().double() * 0.01
    t0 = d[:, :, 0].min()
    st_, en_ = d[:, 0, 0] - t0, d[:, 0, 14] - t0
    print(f"{name}: launch {sorted(t)[2]:6.1f} us; block starts median {st_.median():6.2f} us, last {st_.max():6.2f}; block duration "
          f"mean {(en_ - st_).mean():6.2f} max {(en_ - st_).max():6.2f}; last end {en_.max():6.2f} us")
    prev = None
    for k in ORDER[name]:
        v = d[:, :, k] - t0
        print(f"  {k:2d} {NAMES[k]:36s} w0 {v[:, 0].mean():7.2f}  w3 {v[:, 3].mean():7.2f}  max {v.max():7.2f}"
              + ("" if prev is None else f"   step w0 {(d[:, 0, k] - d[:, 0, prev]).mean():6.2f}  w3 {(d[:, 3, k] - d[:, 3, prev]).mean():6.2f}"))
        prev = k
