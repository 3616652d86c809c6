#!/usr/bin/env python3
"""Time the f32-MFMA and bf16x3-MFMA NT kernels on the SwinIR training shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

T = 32768
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (N, K, a_mode, epi) in [(540, 180, 1, 0), (180, 180, 0, 2), (360, 180, 1, 0), (180, 360, 2, 2),
                            (360, 180, 0, 3), (180, 540, 0, 0)]:
    A = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    R = torch.randn(T, N, device=dev); out = torch.empty(T, N, device=dev)
    st = torch.stack([A.mean(1), 1 / torch.sqrt(A.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    Wb = ops.split_bf16x3(W)
    kw = dict(out=out, a_mode=a_mode, ln_stats=st if a_mode == 1 else None, epi=epi, R=R if epi >= 2 else None)
    t32 = timeit(lambda: ops.gemm_nt(A, W, b, **kw))
    o32 = out.clone()
    tbx = timeit(lambda: ops.gemm_nt(A, Wb, b, **kw))
    fl = 2.0 * T * N * K
    print(f"N={N:4d} K={K:4d} a={a_mode} epi={epi}: f32 {t32:7.1f} us ({fl/t32*1e-6:6.1f} TF/s)   "
          f"bx3 {tbx:7.1f} us ({fl/tbx*1e-6:6.1f} TF/s)   maxdiff {(out-o32).abs().max().item():.2e}")

x = torch.randn(8, 64, 64, 180, device=dev)
wp = torch.randn(9, 180, 180, device=dev) * 0.02
b = torch.randn(180, device=dev)
y = torch.empty(8, 64, 64, 180, device=dev)
wb = ops.split_bf16x3(wp)
t32 = timeit(lambda: ops.conv3x3(x, wp, b, 180, out=y)); y32 = y.clone()
tbx = timeit(lambda: ops.conv3x3(x, wb, b, 180, out=y))
fl = 2.0 * 8 * 64 * 64 * 180 * 180 * 9
print(f"conv 180->180 B=8 64x64: f32 {t32:7.1f} us ({fl/t32*1e-6:6.1f} TF/s)   bx3 {tbx:7.1f} us "
      f"({fl/tbx*1e-6:6.1f} TF/s)   maxdiff {(y-y32).abs().max().item():.2e}")
W = torch.randn(540, 180, device=dev)
bx = ops.Bx3(540, 180, W.device)
print(f"split 540x180: {timeit(lambda: bx.fill(W)):.1f} us")

# weight gradients: grouped TN (the four Linear layers of a block)
import os
C, hid = 180, 360
g = torch.randn(T, C, device=dev); dqkv = torch.randn(T, 3 * C, device=dev); dh = torch.randn(T, hid, device=dev)
x = torch.randn(T, C, device=dev); hb = torch.randn(T, hid, device=dev)
st = torch.stack([x.mean(1), 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
def mk():
    return [dict(dY=g, X=hb, dW=torch.empty(C, hid, device=dev), db=torch.empty(C, device=dev), b_mode=2),
            dict(dY=dh, X=x, dW=torch.empty(hid, C, device=dev), db=torch.empty(hid, device=dev), b_mode=1, ln_stats=st),
            dict(dY=g, X=x, dW=torch.empty(C, C, device=dev), db=torch.empty(C, device=dev)),
            dict(dY=dqkv, X=x, dW=torch.empty(3 * C, C, device=dev), db=torch.empty(3 * C, device=dev), b_mode=1, ln_stats=st)]
res = {}
for mode in ("f32", "bx3"):
    os.environ["SRHIP_MM"] = mode
    pr = mk()
    t = timeit(lambda: ops.linear_wgrad_grouped(pr))
    res[mode] = (t, [q["dW"].clone() for q in pr], [q["db"].clone() for q in pr])
fl = 2.0 * T * (C * hid * 2 + C * C + 3 * C * C)
d = max((a - b).abs().max().item() / a.abs().max().item() for a, b in zip(res["f32"][1], res["bx3"][1]))
d2 = max((a - b).abs().max().item() / a.abs().max().item() for a, b in zip(res["f32"][2], res["bx3"][2]))
print(f"grouped wgrad (+reducers): f32 {res['f32'][0]:7.1f} us ({fl/res['f32'][0]*1e-6:6.1f} TF/s)   bx3 {res['bx3'][0]:7.1f} us "
      f"({fl/res['bx3'][0]*1e-6:6.1f} TF/s)   rel maxdiff dW {d:.2e} db {d2:.2e}")
dy = torch.randn(8, 64, 64, 180, device=dev); xx = torch.randn(8, 64, 64, 180, device=dev)
for mode in ("f32", "bx3"):
    os.environ["SRHIP_MM"] = mode
    dW = torch.empty(180, 180, 3, 3, device=dev); db = torch.empty(180, device=dev)
    t = timeit(lambda: ops.conv3x3_wgrad(dy, xx, dW, db))
    res[mode] = (t, dW.clone(), db.clone())
fl = 2.0 * 8 * 64 * 64 * 180 * 180 * 9
print(f"conv wgrad (+reducer): f32 {res['f32'][0]:7.1f} us ({fl/res['f32'][0]*1e-6:6.1f} TF/s)   bx3 {res['bx3'][0]:7.1f} us "
      f"({fl/res['bx3'][0]*1e-6:6.1f} TF/s)   rel maxdiff {((res['f32'][1]-res['bx3'][1]).abs().max()/res['f32'][1].abs().max()).item():.2e}")
