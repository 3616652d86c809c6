#!/usr/bin/env python3
"""Wall-clock stamps (100 MHz) of the phases of the fused MLP kernels (mlp_f16.hip), all waves of all blocks.  Needs an
experiment build (make -C sr-caco-2_amd/csrc EXPERIMENTS=1 OUT=../lib/libsrhip_exp.so OBJDIR=../lib/obj_exp) loaded
through SRHIP_LIB."""
import os, sys, ctypes
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
h = torch.empty(M, hid, device=dev); out = torch.empty(M, C, device=dev); sto = torch.empty(M, 2, device=dev)
dh = torch.empty(M, hid, device=dev); gh = torch.empty(M, hid, device=dev); dx = torch.empty(M, C, device=dev)
nblk = M // 64
dbg = torch.zeros(nblk, 4, 16, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_mlp_debug_buffer
fn.argtypes = [ctypes.c_void_p]

def fwd(): ops.mlp_fwd_f16(x, st, P["w1"], b1f, P["w2"], b2, out, h=h, stats_out=sto)
def bwd(): ops.mlp_bwd_f16(dy, P["w2T"], P["w1T"], h, dh, gh, x, st, dx)
names = ["start", "x staged", "barrier", "gemm1 done", "activation done", "max+barrier", "h pass 1 staged", "barrier",
         "gemm2 pass 1", "barrier", "h pass 2 staged+barrier", "gemm2 pass 2", "barrier", "T laid+barrier", "end"]
for name, f in (("forward", fwd), ("backward", bwd)):
    for _ in range(3):
        f()
    fn(dbg.data_ptr()); f(); torch.cuda.synchronize(); fn(None)
    d = dbg.cpu().double() * 0.01
    t0 = d[:, :, 0].min()
    st_, en_ = d[:, 0, 0] - t0, d[:, 0, 14] - t0
    print(f"{name}: block starts median {st_.median():6.2f} us, last {st_.max():6.2f}; block duration mean {(en_ - st_).mean():6.2f} "
          f"max {(en_ - st_).max():6.2f}; last end {en_.max():6.2f} us")
    for k, nme in enumerate(names):
        v = d[:, :, k] - t0
        print(f"  {k:2d} {nme:24s} w0 {v[:, 0].mean():7.2f}  w3 {v[:, 3].mean():7.2f}  max {v.max():7.2f}"
              + ("" if k == 0 else f"   step w0 {(d[:, 0, k] - d[:, 0, k - 1]).mean():6.2f}  w3 {(d[:, 3, k] - d[:, 3, k - 1]).mean():6.2f}"))
