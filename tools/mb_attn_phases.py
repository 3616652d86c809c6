#!/usr/bin/env python3
"""Wall-clock stamps (100 MHz) inside the fp16x2 attention backward (k_wattn2_bwd), all waves of all blocks.  Needs an
experiment build loaded through SRHIP_LIB (see mb_wmsa_phases.py)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch
from srhip import ops

B, H, W, C, heads = 8, 64, 64, 180, 6
T = B * H * W
dev = "cuda"
qkv = torch.randn(T, 3 * C, device=dev); dout = torch.randn(T, C, device=dev); dqkv = torch.empty(T, 3 * C, device=dev)
table = torch.randn(225, heads, device=dev) * 0.5
biasF, biasG = torch.empty(heads, 64, 64, device=dev), torch.empty(heads, 64, 64, device=dev)
ops.bias_expand_f16(table, biasF, biasG)
dbT = torch.empty(heads, 64, 64, device=dev)
nblk = (B * (H // 8) * (W // 8) + 3) // 4 * heads
dbg = torch.zeros(nblk, 4, 16, dtype=torch.int64, device=dev)
fn = ops.lib.srhip_wattn2_debug_buffer
fn.argtypes = [ctypes.c_void_p]
names = ["start", "K,V rows split", "K^T gathered", "I=0", "I=1", "I=2", "I=3", "Q,dO rows split", "Q^T,dO^T gathered",
         "J=0", "J=1", "J=2", "J=3", "end (partial tile)"]
for shift in (0, 4):
    for _ in range(3):
        ops.window_attention_bwd_f16(qkv, dout, dqkv, biasF, biasG, dbT, B, H, W, C, heads, shift)
    fn(dbg.data_ptr())
    ops.window_attention_bwd_f16(qkv, dout, dqkv, biasF, biasG, dbT, B, H, W, C, heads, shift)
    torch.cuda.synchronize(); fn(None)
    d = dbg.cpu().double() * 0.01
    t0 = d[:, :, 0].min()
    st_, en_ = d[:, 0, 0] - t0, d[:, 0, 13] - t0
    first = st_ < 5.0
    print(f"shift {shift}: {nblk} blocks; {int(first.sum())} start within 5 us, the others at median {st_[~first].median():6.2f} us "
          f"(last {st_.max():6.2f}); duration first wave {(en_ - st_)[first].mean():6.2f} us, later {(en_ - st_)[~first].mean():6.2f} us; "
          f"last end {en_.max():6.2f} us")
    for grp, nm in ((first, "first-round blocks"), (~first, "later blocks")):
        dd = d[grp]
        print(f"  {nm}: step durations (us, wave 0 mean): " + "  ".join(
            f"{names[k]} {(dd[:, 0, k] - dd[:, 0, k - 1]).mean():5.2f}" for k in range(1, 14)))
