#!/usr/bin/env python3
"""Cycle stamps of the phases of the fused W-MSA forward (wmsa_f16.hip), all waves of all blocks.  Needs an experiment
build: make -C sr-caco-2_amd/csrc EXPERIMENTS=1 OUT=../lib/libsrhip_exp.so OBJDIR=../lib/obj_exp, run with
SRHIP_LIB=sr-caco-2_amd/lib/libsrhip_exp.so."""
import os, sys, ctypes
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
qkv = None if os.environ.get('MB_NO_QKV') else torch.empty(T, 3 * C, device=dev); att = torch.empty(T, C, device=dev); out = torch.empty(T, C, device=dev)
sto = torch.empty(T, 2, device=dev)
nblk = T // 64
NW = int(os.environ.get("SRHIP_WMSA_NW", "6"))
dbg = torch.zeros(nblk, NW, 16, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_wmsa_debug_buffer
fn.argtypes = [ctypes.c_void_p]
for shift in (0, 4):
    for _ in range(3):
        ops.wmsa_fwd_f16(x, st, Pq, bqf, Pp, bp, biasF, qkv, att, out, B, H, W, heads, shift, stats_out=sto)
    if os.environ.get("MB_COLD", "1") != "0":      # as in the training step: nothing of the launch's operands in the Infinity Cache
        big = torch.empty(1 << 28, device=dev)     # 1 GiB
        big.fill_(1.0); big.mul_(2.0)
        torch.cuda.synchronize()
    fn(dbg.data_ptr())
    ops.wmsa_fwd_f16(x, st, Pq, bqf, Pp, bp, biasF, qkv, att, out, B, H, W, heads, shift, stats_out=sto)
    torch.cuda.synchronize()
    fn(None)
    d = dbg.cpu().double()
    d = d * 0.01                  # 100 MHz wall clock -> microseconds
    t0 = d[:, :, 0].min()
    st_, en_ = d[:, 0, 0] - t0, d[:, 0, 9] - t0
    print(f"  block starts: median {st_.median():6.2f} us, last {st_.max():6.2f} us; block duration mean {(en_ - st_).mean():6.2f} "
          f"max {(en_ - st_).max():6.2f}; last end {en_.max():6.2f} us")
    names = ["start", "x staged", "barrier", "qkv done", "barrier", "attention done", "barrier", "a staged+barrier",
             "proj mfma done", "end"]
    print(f"shift {shift}: microseconds since the first wave's start (mean over blocks; wave 0 / wave {NW - 1}; max over all waves)")
    q = [2, 10, 11, 12, 13, 14, 3]
    print("  qkv phase, wave 0 mean: " + "  ".join(f"{n} {(d[:, 0, b] - d[:, 0, a]).mean():6.2f}" for n, a, b in
          zip(["mfma q", "store q", "mfma k", "store k", "mfma v", "store v"], q[:-1], q[1:])))
    for k, nme in enumerate(names):
        v = d[:, :, k] - t0
        print(f"  {k} {nme:18s} w0 {v[:, 0].mean():8.2f}  w3 {v[:, 3].mean():8.2f}  max {v[:, :4].max():8.2f}"
              + ("" if k == 0 else f"   step (w0 mean) {(d[:, 0, k] - d[:, 0, k - 1]).mean():7.2f}  (w3) {(d[:, 3, k] - d[:, 3, k - 1]).mean():7.2f}"))
