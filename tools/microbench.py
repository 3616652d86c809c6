#!/usr/bin/env python3
"""Per-kernel timing on the GPU box (HIP events, many iterations)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sr-caco-2_amd"))
import torch  # noqa: E402
from srhip import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    which = sys.argv[1:] or ["nt", "tn", "attn", "conv"]
    T = 32768
    dev = "cuda"
    if "nt" in which:
        for (N, K) in [(180, 180), (360, 180), (540, 180), (180, 360), (180, 540)]:
            A = torch.randn(T, K, device=dev)
            W = torch.randn(N, K, device=dev) * 0.1
            b = torch.randn(N, device=dev)
            R = torch.randn(T, N, device=dev)
            out = torch.empty(T, N, device=dev)
            st = torch.randn(T, 2, device=dev)
            for name, kw in [("bias", {}), ("ln", dict(a_mode=1, ln_stats=st)),
                             ("gelu+res", dict(a_mode=2, epi=2, R=R)), ("dgelu", dict(epi=3, R=R))]:
                us = timeit(lambda: ops.gemm_nt(A, W, b, out=out, **kw))
                print(f"nt M={T} N={N} K={K} {name:9s} {us:8.1f} us  {2 * T * N * K / us / 1e6:6.1f} TF/s",
                      flush=True)
    if "tn" in which:
        for (NI, NJ) in [(180, 180), (540, 180), (360, 180), (180, 360)]:
            dY = torch.randn(T, NI, device=dev)
            X = torch.randn(T, NJ, device=dev)
            dW = torch.empty(NI, NJ, device=dev)
            db = torch.empty(NI, device=dev)
            us = timeit(lambda: ops.linear_wgrad(dY, X, dW, db))
            print(f"tn M={T} NI={NI} NJ={NJ} {us:8.1f} us  {2 * T * NI * NJ / us / 1e6:6.1f} TF/s", flush=True)
    if "attn" in which:
        B, H, W, C, heads = 8, 64, 64, 180, 6
        qkv = torch.randn(T, 3 * C, device=dev)
        out = torch.empty(T, C, device=dev)
        dout = torch.randn(T, C, device=dev)
        dqkv = torch.empty(T, 3 * C, device=dev)
        bT = torch.randn(heads, 64, 64, device=dev)
        bN = bT.transpose(1, 2).contiguous()
        dbT = torch.zeros(heads, 64, 64, device=dev)
        for shift in (0, 4):
            us = timeit(lambda: ops.window_attention_fwd(qkv, out, bT, B, H, W, C, heads, shift))
            fl = 4 * 64 * 64 * 30 * 512 * 6
            print(f"attn fwd shift={shift} {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)
            us = timeit(lambda: ops.window_attention_bwd(qkv, dout, dqkv, bT, bN, dbT, B, H, W, C, heads, shift))
            print(f"attn bwd shift={shift} {us:8.1f} us  {2.5 * fl / us / 1e6:6.1f} TF/s (useful)", flush=True)
    if "conv" in which:
        B, H, W, C = 8, 64, 64, 180
        x = torch.randn(B, H, W, C, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        wp = torch.empty(9, C, C, device=dev)
        ops.pack_conv_weight(w, wp, None)
        bias = torch.randn(C, device=dev)
        out = torch.empty(B, H, W, C, device=dev)
        us = timeit(lambda: ops.conv3x3(x, wp, bias, C, out=out))
        fl = 2 * B * H * W * C * C * 9
        print(f"conv3x3 {C}->{C} {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)
        dw = torch.empty_like(w)
        db = torch.empty(C, device=dev)
        us = timeit(lambda: ops.conv3x3_wgrad(out, x, dw, db))
        print(f"conv3x3 wgrad {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
