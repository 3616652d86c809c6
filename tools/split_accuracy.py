#!/usr/bin/env python3
"""Accuracy of the candidate operand splits for an f32-accurate GEMM on the matrix core, emulated on the CPU (no GPU):
products of the split parts are exact in f32 (8+8 / 11+11 significant bits), the sums run in f32 as in the MFMA.
  bf16x3 : x = h + m + l (bf16 each), six products                      -- what libsrhip runs
  fp16x2 : x' = x * 2^s (power of two, per tensor or per row), x' = h + l (fp16 each), three products (h.h + h.l + l.h)
  f32    : plain f32 matmul
Operands: a LayerNorm output (rows of rms 1) against trained-like Linear weights (trunc-normal 0.02 x 10, as the g18
goldens), and a heavy-tailed pair (activations x 1e-3 .. 1e3, the case a per-tensor scale does not cover).
    python tools/split_accuracy.py"""
import torch

torch.manual_seed(0)


def split_bf16x3(x):
    h = x.bfloat16().float(); r = x - h
    m = r.bfloat16().float(); l = (r - m).bfloat16().float()
    return h, m, l


def split_fp16x2(x, per_row=False):
    mx = x.abs().amax(1, keepdim=True).clamp_min(1e-30) if per_row else x.abs().max()
    s = torch.floor(torch.log2(16384.0 / mx))
    xs = x * 2.0 ** s
    h = xs.half().float()
    l = (xs - h).half().float()
    return h, l, 2.0 ** (-s)


def report(name, A, W):
    ref = A.double() @ W.double().t()
    den = ref.abs().max()
    f32 = (A @ W.t()).double()
    ah, am, al = split_bf16x3(A); wh, wm, wl = split_bf16x3(W)
    b3 = (am @ wm.t() + ah @ wl.t() + al @ wh.t() + ah @ wm.t() + am @ wh.t() + ah @ wh.t()).double()
    ah, al, sa = split_fp16x2(A); wh, wl, sw = split_fp16x2(W)
    f2 = ((ah @ wl.t() + al @ wh.t() + ah @ wh.t()) * (sa * sw)).double()
    ah, al, sa = split_fp16x2(A, True); wh, wl, sw = split_fp16x2(W, True)       # per-row block exponents (exact scaling)
    f2r = ((ah @ wl.t() + al @ wh.t() + ah @ wh.t()) * sa * sw.t()).double()
    e = lambda y: ((y - ref).abs().max() / den).item()
    r = lambda y: ((y - ref).norm() / ref.norm()).item()
    rr = lambda y: ((y - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300)).max().item()      # worst ROW, relative to that row
    print(f"{name:34s} L2-rel  f32 {r(f32):.2e}  bf16x3 {r(b3):.2e}  fp16x2 {r(f2):.2e}  fp16x2/row {r(f2r):.2e}   |  worst row  f32 {rr(f32):.2e}  bf16x3 {rr(b3):.2e}  fp16x2 {rr(f2):.2e}  fp16x2/row {rr(f2r):.2e}")


M, K, N = 4096, 180, 540
x = torch.randn(M, K) * torch.rand(M, 1) * 3 + torch.randn(M, 1)
A = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
W = torch.nn.init.trunc_normal_(torch.empty(N, K), std=0.02) * 10
report("LayerNorm output x Linear weight", A, W)
report("gelu(h) x Linear weight (K 360)", torch.nn.functional.gelu(torch.randn(M, 360) * 2), torch.nn.init.trunc_normal_(torch.empty(180, 360), std=0.02) * 10)
report("gradient-like (1e-7 scale) x W^T", torch.randn(M, 540) * 1e-7 * torch.exp(torch.randn(M, 1) * 2), W.t().contiguous())
report("heavy-tailed rows (1e-3 .. 1e3)", torch.randn(M, K) * torch.exp(torch.randn(M, 1) * 3), W)


# ---- 3x3 conv (implicit GEMM): the accumulator of an output pixel mixes nine neighbouring pixels, so the activation's block
#      exponent cannot be per pixel -- the candidate is ONE power-of-two scale per halo tile (8 x 16 output pixels + halo, all
#      channels), weights per output channel.  Emulated tile by tile; errors per output pixel relative to that pixel's own norm.
def conv_report(name, x, w):
    import torch.nn.functional as F
    B, C, H, W = x.shape
    ref = F.conv2d(x.double(), w.double(), padding=1)
    f32 = F.conv2d(x, w, padding=1).double()
    co = w.shape[0]
    ws = torch.floor(torch.log2(16384.0 / w.abs().amax((1, 2, 3), keepdim=True)))
    wsc = w * 2.0 ** ws
    wh = wsc.half().float(); wl = (wsc - wh).half().float()
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros(B, co, H, W, dtype=torch.float64)
    for b in range(B):
        for y0 in range(0, H, 8):
            for x0 in range(0, W, 16):
                t = xp[b:b + 1, :, y0:y0 + 10, x0:x0 + 18]
                mx = t.abs().max()
                s = torch.floor(torch.log2(16384.0 / mx)) if mx > 0 else torch.tensor(0.0)
                ts = t * 2.0 ** s
                th = ts.half().float(); tl = (ts - th).half().float()
                y = F.conv2d(th, wl) + F.conv2d(tl, wh) + F.conv2d(th, wh)
                out[b, :, y0:y0 + 8, x0:x0 + 16] = (y * 2.0 ** (-s)).double()[0] * (2.0 ** (-ws)).double().view(co, 1, 1)
    pix = lambda y: ((y - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-300))
    r = lambda y: ((y - ref).norm() / ref.norm()).item()
    print(f"{name:40s} L2-rel  f32 {r(f32):.2e}  fp16x2/tile {r(out):.2e}   |  worst pixel  f32 {pix(f32).max().item():.2e}  fp16x2/tile {pix(out).max().item():.2e}"
          f"   99.9 % pixel  f32 {pix(f32).flatten().kthvalue(int(0.999 * pix(f32).numel())).values.item():.2e}  fp16x2/tile {pix(out).flatten().kthvalue(int(0.999 * pix(out).numel())).values.item():.2e}")


torch.manual_seed(1)
w = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
feat = torch.nn.functional.relu(torch.randn(1, 64, 64, 64)) * torch.exp(torch.nn.functional.interpolate(torch.randn(1, 1, 8, 8), size=(64, 64), mode="bicubic") * 1.5)
conv_report("conv 64->64: ReLU features, smooth gain", feat, w)
grad = torch.randn(1, 64, 64, 64) * 1e-7 * torch.exp(torch.randn(1, 1, 64, 64) * 2.5)
conv_report("conv 64->64: gradient-like, per-pixel e^2.5N", grad, w)
