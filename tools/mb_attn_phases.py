#!/usr/bin/env python3
"""Wall-clock stamps (100 MHz) inside the fp16x2 attention backward (k_wattn3_bwd): every wave of every block, per
window of the block.  Needs an experiment build loaded through SRHIP_LIB (make EXPERIMENTS=1 OUT=../lib/libsrhip_exp.so
OBJDIR=../lib/obj_exp)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
qs = [torch.randn(T, 3 * C, device=dev) for _ in range(4)]
dout = torch.randn(T, C, device=dev); dqkv = torch.empty(T, 3 * C, device=dev)
table = torch.randn(225, heads, device=dev) * 0.5
biasF, biasG = torch.empty(heads, 64, 64, device=dev), torch.empty(heads, 64, 64, device=dev)
ops.bias_expand_f16(table, biasF, biasG)
parts = torch.empty(ops.wattn_dbias_ws(B, H, W, heads), device=dev)
nblk = (B * (H // 8) * (W // 8) + 3) // 4 * heads
dbg = torch.zeros(nblk, 4, 4, 8, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_wattn2_debug_buffer
fn.argtypes = [ctypes.c_void_p]
names = ["rows -> images", "barrier", "query side", "barrier", "key side", "barrier"]
for shift in (0, 4):
    for i in range(3):
        ops.window_attention_bwd_f16(qs[i], dout, dqkv, biasF, biasG, None, B, H, W, C, heads, shift, parts=parts)
    fn(dbg.data_ptr())
    ops.window_attention_bwd_f16(qs[3], dout, dqkv, biasF, biasG, None, B, H, W, C, heads, shift, parts=parts)
    torch.cuda.synchronize(); fn(None)
    d = dbg.cpu().double() * 0.01
    t0 = d[:, :, 0, 0].min()
    st_, en_ = d[:, 0, 0, 0] - t0, d[:, :, 3, 6].max(1).values - t0
    print(f"shift {shift}: {nblk} blocks; starts {st_.min():.2f} .. {st_.median():.2f} (median) .. {st_.max():.2f} us; "
          f"block duration mean {(en_ - st_).mean():.2f} us; last end {en_.max():.2f} us")
    for wi in range(4):
        print(f"  window {wi}: " + "  ".join(f"{names[k]} {(d[:, :, wi, k + 1] - d[:, :, wi, k]).mean():5.2f}" for k in range(6))
              + f"   | top of next - end {((d[:, :, wi + 1, 0] - d[:, :, wi, 6]).mean() if wi < 3 else 0):5.2f}")
