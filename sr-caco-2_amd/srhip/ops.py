"""Thin tensor-level wrappers over the libsrhip C-ABI (include/srhip.h).

PyTorch is used for device memory and streams only: every function here takes
CUDA(=HIP) tensors, passes raw pointers + sizes + the current stream to the
library, and returns tensors allocated by torch.  No function computes with
aten ops and none has a CPU path."""
import ctypes

import torch

from . import probe
from ._lib import call, lib, SrhipError  # noqa: F401


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise SrhipError("srhip ops need CUDA/HIP tensors (no CPU fallback exists)")
        if t.dtype not in (torch.float32, torch.float64, torch.int32):
            raise SrhipError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rows(t):
    """2-D view geometry of a row-major matrix whose last dim is contiguous."""
    assert t.stride(-1) == 1
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


# ------------------------------------------------------------------ contractions
def use_bx3():
    """Matrix path: bf16x3 split MFMA (default) or exact-f32 MFMA (env SRHIP_MM=f32)."""
    import os
    return os.environ.get("SRHIP_MM", "bx3") != "f32"


import os as _os
# weight-gradient (TN) side: bf16x3 from 64 channels on, once 64-wide tiles got their own plan (three blocks per
# CU) and unequal operand widths the smaller tile class -- 64->64 at 8x128x128: 186 us (f32) vs 148 us
BX3_MIN_CHANNELS = int(_os.environ.get("SRHIP_BX3_MIN_CH", "64"))
# Linear weights whose GEMMs run on 192-column tiles are prepared as TWO fp16 planes with per-row power-of-two scales and
# multiplied with THREE products (k_nth2, gemm_ntw.hip; srhip_gemm_nt_f16x2): f32-grade per row, +6 % on the SwinIR step.
# SRHIP_F16X2=0: three bf16 planes / six products (k_ntw) for those too.
F16X2 = _os.environ.get("SRHIP_F16X2", "1") not in ("", "0")
# ... and the 3x3 convs with 64 / 128 / 192 / 256 (64-column tiles or slices, the PixelShuffle-fused upsampler convs included) or
# 180 (SwinIR: 192-column tiles) output channels and 64 .. 256 input channels: two fp16 planes, one power-of-two scale per
# weight output channel and per activation halo tile, three products (k_nhcw2 / k_nhcw): f32-grade per output pixel.
# SRHIP_F16X2_CONV=0: bf16x3 everywhere; _WIDE=0: only the plain 64-column convs; SRHIP_F16X2_CONV180=0: SwinIR's on k_ntcw.
F16X2_CONV = _os.environ.get("SRHIP_F16X2_CONV", "1") not in ("", "0")
F16X2_CONV_WIDE = _os.environ.get("SRHIP_F16X2_CONV_WIDE", "1") != "0"
F16X2_CONV180 = _os.environ.get("SRHIP_F16X2_CONV180", "1") != "0"
# widest output / reduce side that takes the fp16x2 conv (DBPN / SRFBN's stride-8 transposed convs are 64 -> 4096 and back)
F16X2_CONV_MAX = int(_os.environ.get("SRHIP_F16X2_CONV_MAX", "4096"))


# The NT side (conv / Linear forward and data gradient) has its own threshold: at 64 -> 64 channels, B=8, 128x128
# (tools/mb_conv64.py) the conv goes 108.8 -> 65.5 us on the bf16x3 kernel; the weight-gradient side only caught
# up (186 -> 148 us) after its plan for 64-wide tiles changed -- with the plan of the wide tiles it was slower (212 us).
# Round 4: 32 -- ProSR's DenseNet convs (40- / 80-channel growth) ran on the exact-f32 kernel under 64: same box, x8 / x4 / x2
# evaluation 64.7 -> 53.8 / 112.8 -> 90.1 / 173.1 -> 135.3 ms per batch on the split kernels; 16 changes nothing further.
BX3_MIN_CHANNELS_NT = int(_os.environ.get("SRHIP_BX3_MIN_CH_NT", "32"))


def bx3_nt_for(*channels):
    """bf16x3 kernels for the NT side (forward / data gradient) of this problem?"""
    return use_bx3() and min(channels) >= BX3_MIN_CHANNELS_NT


def bx3_for(*channels):
    """bf16x3 kernels for this problem?  (mode switch + smallest channel count involved)"""
    return use_bx3() and min(channels) >= BX3_MIN_CHANNELS


def set_matmul_mode(mode):
    """0: f32-accurate bf16x3 (default) | 1: single bf16 product, inference only (srhip_set_matmul_mode)."""
    call("srhip_set_matmul_mode", int(mode))


# --amp at evaluation time: the plain conv family (VDSR, DRRN, EDSR) runs on fp16 STORAGE (conv_h16.hip: activations fp16 in
# HBM, one fp16 product); SRHIP_H16_EVAL=0: the f32-storage single-product kernels, as every other net
H16_EVAL = _os.environ.get("SRHIP_H16_EVAL", "1") != "0"


def h16_eval():
    """inside ``amp_inference(True)`` with the fp16-storage path enabled"""
    return H16_EVAL and lib.srhip_get_matmul_mode() == 1


class amp_inference:
    """``with ops.amp_inference(on):`` -- reduced-precision matmuls / convs inside the block (the role of
    torch.cuda.amp.autocast at evaluation time in the reference, model_plain.py:322-327)."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = lib.srhip_get_matmul_mode()
        if self.on:
            set_matmul_mode(1)

    def __exit__(self, *exc):
        set_matmul_mode(self.prev)


def _tn_sfx(bx=None):
    return "_bx3" if (use_bx3() if bx is None else bx) else ""


class Bx3:
    """Weight operand pre-split into three bf16 planes [3][rows][Kp] for the bf16x3
    contractions (srhip_split_bf16x3).  rows = N (Linear) or 9*Cout (conv pack)."""
    __slots__ = ("planes", "rows", "K", "fmt")

    def __init__(self, rows, K, device):
        kp = lib.srhip_bf16x3_kp(K)
        self.planes = torch.empty(3, rows, kp, device=device, dtype=torch.int16)
        self.rows, self.K = rows, K
        self.fmt = 0      # 0: three bf16 planes | 1: two fp16 planes + per-row 2^-s (PrepTable.linear under SRHIP_F16X2)

    def fill(self, W2d):
        """W2d: [rows][K] f32 view (row stride = W2d.stride(0))."""
        assert W2d.shape == (self.rows, self.K) and W2d.stride(1) == 1 and W2d.dtype == torch.float32
        call("srhip_split_bf16x3", _p(W2d), W2d.stride(0), self.rows, self.K, _p(self.planes), _st())
        self.fmt = 0
        return self


def split_bf16x3(W):
    """f32 weight [N,K] or conv pack [9,Cout,Cin] -> Bx3."""
    W2d = W.reshape(-1, W.shape[-1])
    return Bx3(W2d.shape[0], W2d.shape[1], W.device).fill(W2d)


class _PrepEntry(ctypes.Structure):   # srhip_prep_entry (include/srhip.h)
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("c", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("out2", ctypes.c_void_p),
                ("kind", ctypes.c_int), ("blk0", ctypes.c_int),
                ("n0", ctypes.c_int), ("n1", ctypes.c_int), ("n2", ctypes.c_int),
                ("s0", ctypes.c_int), ("s1", ctypes.c_int), ("s2", ctypes.c_int), ("off", ctypes.c_int),
                ("mode", ctypes.c_int)]


class PrepTable:
    """Job table of the per-step weight preparation (srhip_prep_table): every derived
    matmul operand of a network rebuilt by ONE launch.  Holds raw pointers: sources
    and outputs must stay allocated (``keep`` pins them)."""

    def __init__(self):
        self.jobs, self.keep, self.table = [], [], None

    def _add(self, **kw):
        e = _PrepEntry()
        for k, v in kw.items():
            setattr(e, k, v)
        self.jobs.append(e)

    def linear(self, W, out, gamma=None, transpose=False, f16=None):
        """planes of W [N,K] (or W^T), optionally times gamma[k] (LayerNorm fold).  f16: force (True) / forbid
        (False) the two-plane fp16 format; None: the routing rule of the Linear GEMMs."""
        N, K = W.shape
        assert W.is_contiguous() and W.dtype == torch.float32
        rows, kd = (K, N) if transpose else (N, K)
        assert (out.rows, out.K) == (rows, kd)
        self.keep += [W, out, gamma]
        # two fp16 planes + per-row power-of-two scales (prep kind 3) for the operands of the GEMMs that run on
        # 192-column tiles (k_nth2, gemm_ntw.hip) -- same routing rule as sr_gemm_ntp; SRHIP_F16X2=0: bf16x3
        if f16 is None:
            f16 = F16X2 and (rows % 180 == 0 or (rows > 128 and rows % 128 != 0)) and kd <= 1024
        out.fmt = 1 if f16 else 0
        self._add(kind=3 if f16 else 0, a=_p(W), b=_p(gamma), out=_p(out.planes), n0=rows, n1=1, n2=kd, s0=0,
                  s1=1 if transpose else K, s2=K if transpose else 1, off=0,
                  mode=0 if gamma is None else (2 if transpose else 1))

    def conv(self, w, out, data_grad=False, ps2=False, force_f16=False):
        """planes of the tap-major pack [9][Co][Ci] of a conv weight [Co,Ci,3,3], or of the
        flipped / transposed twin [9][Ci][Co] used by the data gradient.  ps2: the conv feeds a
        PixelShuffle(2) fused into the kernel (conv3x3_ps2*): its output channels in sub-pixel-major
        order sp*(Co/4) + c <- torch channel c*4 + sp."""
        Co, Ci = w.shape[:2]
        assert w.is_contiguous() and tuple(w.shape[2:]) == (3, 3) and (not ps2 or Co % 4 == 0)
        rows, kd = (Ci, Co) if data_grad else (Co, Ci)
        assert (out.rows, out.K) == (9 * rows, kd)
        self.keep += [w, out]
        # two fp16 planes + per-output-channel scales (prep kind 4; SRHIP_F16X2_CONV=0: bf16x3) for the convs that run
        # on 64-column tiles / slices (k_nhcw2, gemm_ntw.hip): output side a multiple of 64 up to 256, reduce side 64 .. 256
        f16 = F16X2_CONV and rows % 64 == 0 and rows <= F16X2_CONV_MAX and 64 <= kd <= F16X2_CONV_MAX
        if F16X2_CONV and F16X2_CONV180 and not ps2 and rows % 180 == 0 and 64 <= kd <= 256:
            f16 = True              # SwinIR's 180-column convs: k_nhcw (64-pixel x 192-column tiles); SRHIP_F16X2_CONV180=0: k_ntcw
        if not F16X2_CONV_WIDE:     # SRHIP_F16X2_CONV_WIDE=0: only the 64-column convs without a fused PixelShuffle
            f16 = f16 and rows == 64 and not ps2
        if force_f16:               # operands of the fp16-storage kernels (conv_h16.hip) whatever the f32-grade kernel would take
            assert rows % 64 == 0 and kd % 32 == 0 and rows <= 4096 and kd <= 4096
            f16 = True
        out.fmt = 1 if f16 else 0
        kind = 4 if f16 else 0
        if data_grad:   # out[t][ci][co] = w[co][ci][8 - t]
            self._add(kind=kind, a=_p(w), out=_p(out.planes), n0=Ci, n1=9, n2=Co, s0=-1, s1=9, s2=Ci * 9, off=8,
                      mode=16 if ps2 else 0)
        else:           # out[t][co][ci] = w[co][ci][t]
            self._add(kind=kind, a=_p(w), out=_p(out.planes), n0=Co, n1=9, n2=Ci, s0=1, s1=Ci * 9, s2=9, off=0,
                      mode=12 if ps2 else 0)

    def fold_bias(self, W, b, beta, out):
        N, K = W.shape
        self.keep += [W, b, beta, out]
        self._add(kind=1, a=_p(W), b=_p(b), c=_p(beta), out=_p(out), n0=N, n1=K)

    def bias_expand(self, table, biasT, biasN, heads, biasF=None, biasG=None):
        """dense relative-position bias images: biasT / biasN in the 32x32 accumulator orders of wattn.hip, biasF /
        biasG (optional, together) in the S^T / S orders of the fp16x2 attention kernels (wattn2.hip)."""
        self.keep += [table, biasT, biasN, biasF, biasG]
        self._add(kind=2, a=_p(table), out=_p(biasT), out2=_p(biasN), c=_p(biasF), b=_p(biasG), n0=heads)

    def build(self, device):
        n = len(self.jobs)
        arr = (_PrepEntry * n)()
        blk = 0
        for i, e in enumerate(self.jobs):
            e.blk0 = blk
            nb = lib.srhip_prep_blocks(ctypes.addressof(e))
            if nb <= 0:
                raise SrhipError(f"srhip_prep_blocks: job {i} (kind {e.kind}) rejected: {lib.srhip_last_error().decode()}")
            blk += nb
            arr[i] = e
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.n, self.blocks = n, blk
        return self

    def run(self):
        call("srhip_prep_table", _p(self.table), self.n, self.blocks, _st())


class WeightSet:
    """Operand form of every matmul / conv weight of a network, by key: the f32 tensor
    (env SRHIP_MM=f32: exact-f32 MFMA kernels) or its Bx3 planes (default: bf16x3 split
    MFMA kernels)."""

    def __init__(self):
        self.use_bx3 = use_bx3()
        self.d = {}

    def register(self, key, t):
        self.d[key] = t

    def planes(self, key, rows, K, device):
        bx = self.d.get(key)
        if not isinstance(bx, Bx3) or (bx.rows, bx.K) != (rows, K):
            bx = self.d[key] = Bx3(rows, K, device)
        return bx

    def __getitem__(self, key):
        return self.d[key]


def gemm_nt(A, W, bias=None, out=None, a_mode=0, ln_stats=None, epi=0, R=None,
            rowscale=None, rows_per_scale=1, alpha=1.0, aux=None, stats_out=None):
    """out[M,N] = epi(pro(A)[M,K] . W[N,K]^T + bias).  W: f32 tensor (exact-f32
    MFMA) or Bx3 (3-way bf16 split MFMA)."""
    bx = isinstance(W, Bx3)
    _chk(A, None if bx else W, bias, out, ln_stats, R, rowscale, aux)
    M, K = A.shape
    N = W.rows if bx else W.shape[0]
    assert (W.K if bx else W.shape[1]) == K
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    tail = (_p(bias), _p(out), out.stride(0), M, N, K,
            a_mode, _p(ln_stats), epi, _p(R), 0 if R is None else R.stride(0), _p(rowscale),
            rows_per_scale, float(alpha), _p(aux), 0 if aux is None else aux.stride(0), _st())
    if bx:
        name = "srhip_gemm_nt_f16x2" if W.fmt == 1 else "srhip_gemm_nt_bx3"
        args = (_p(A), A.stride(0), _p(W.planes)) + tail[:-1] + (_p(stats_out), tail[-1])
    else:
        assert stats_out is None, "row statistics are produced by the bx3 kernel only"
        name, args = "srhip_gemm_nt", (_p(A), A.stride(0), _p(W), W.stride(0)) + tail
    kind = "gemm_nt" if bx else "gemm_nt_f32"
    if probe.on(kind):
        with probe.timed((kind, M, N, K), 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)):
            call(name, *args)
    else:
        call(name, *args)
    return out


def mm(a, b, ta=False, tb=False, bias=None):
    """op(a) . op(b) (+ bias) for the SMALL dense products of the tape nets (squeeze-excitation gates, continuous-position-
    bias MLPs: a few rows each) on the library's own exact-f32 MFMA kernel (srhip_gemm_nt) instead of aten's `@`
    (rocBLAS): op(a) [M, K] and op(b)^T [N, K] are made contiguous, K zero-padded to a multiple of 4."""
    A = a.t() if ta else a
    W = b if tb else b.t()               # srhip_gemm_nt takes W [N, K] and computes A W^T
    K = A.shape[1]
    assert W.shape[1] == K, (A.shape, W.shape)
    if K % 4:
        K4 = (K + 3) // 4 * 4
        Ap = A.new_zeros(A.shape[0], K4); Ap[:, :K] = A
        Wp = W.new_zeros(W.shape[0], K4); Wp[:, :K] = W
        A, W = Ap, Wp
    return gemm_nt(A.contiguous().float(), W.contiguous().float(), None if bias is None else bias.contiguous())


def gemm_nt_batched(A, a_z, W, w_z, C, c_z, M, N, K, zcount, zdiv):
    """zcount products C_z[M, N] = A_z[M, K] . W_z[N, K]^T in one launch (exact-f32 kernel).  A / W / C: 2-D views giving the
    base pointer and row pitch of problem 0; *_z = (stride per z // zdiv, stride per z % zdiv) in floats."""
    _chk(A, W, C)
    assert A.stride(-1) == 1 and W.stride(-1) == 1 and C.stride(-1) == 1
    args = (A.data_ptr(), A.stride(-2), a_z[0], a_z[1], W.data_ptr(), W.stride(-2), w_z[0], w_z[1],
            C.data_ptr(), C.stride(-2), c_z[0], c_z[1], M, N, K, zcount, zdiv, _st())
    if probe.on("gemm_batched"):
        with probe.timed(("gemm_batched", M, N, K, zcount), 2.0 * M * N * K * zcount, 4.0 * zcount * (M * K + N * K + M * N)):
            call("srhip_gemm_nt_batched", *args)
    else:
        call("srhip_gemm_nt_batched", *args)
    return C


def gemm_nt_lnbwd(A, W, x, stats, res, out):
    """out = res + LayerNorm_backward(A . W^T; x, stats) in one kernel (W: Bx3)."""
    assert isinstance(W, Bx3)
    _chk(A, x, stats, res, out)
    M, K = A.shape
    assert W.K == K and x.shape == (M, W.rows) and out.shape == (M, W.rows)
    args = (_p(A), A.stride(0), _p(W.planes), _p(out), out.stride(0), M, W.rows, K,
            _p(x), x.stride(0), _p(stats), _p(res), 0 if res is None else res.stride(0), _st())
    name = "srhip_gemm_nt_f16x2_lnbwd" if W.fmt == 1 else "srhip_gemm_nt_bx3_lnbwd"
    if probe.on("gemm_nt"):      # same kernel as gemm_nt: belongs to the same roofline entry
        with probe.timed(("gemm_nt", M, W.rows, K), 2.0 * M * W.rows * K, 4.0 * (M * K + W.rows * K + 3 * M * W.rows)):
            call(name, *args)
    else:
        call(name, *args)
    return out


def l2norm_rows_(x, k=1.0, eps=5e-5):
    """x[t] <- k * x[t] / max(|x[t]|, eps) on token rows [T, C] (F.normalize over channels, network_enlcn.py:341-342)."""
    _chk(x)
    assert x.dim() == 2 and x.stride(1) == 1
    call("srhip_l2norm_rows", _p(x), x.stride(0), x.shape[0], x.shape[1], float(eps), float(k), _st())
    return x


def l2norm_rows_train_(x, factors, k=1.0, eps=5e-5):
    """l2norm_rows_ that also leaves factors[t] = k / max(|x[t]|, eps) for l2norm_rows_bwd_."""
    _chk(x, factors)
    assert x.dim() == 2 and x.stride(1) == 1 and factors.numel() == x.shape[0]
    call("srhip_l2norm_rows_train", _p(x), x.stride(0), x.shape[0], x.shape[1], float(eps), float(k), _p(factors), _st())
    return x


def l2norm_rows_bwd_(dy, y, factors, k=1.0, eps=5e-5):
    """dy <- the gradient with respect to the un-normalised rows (y = the normalised rows, factors from the forward)."""
    _chk(dy, y, factors)
    assert dy.shape == y.shape and dy.stride(1) == y.stride(1) == 1
    call("srhip_l2norm_rows_bwd", _p(dy), dy.stride(0), _p(y), y.stride(0), _p(factors), dy.shape[0], dy.shape[1], float(eps),
         float(k), _st())
    return dy


def performer_features_bwd_(g, f, eps=1e-4):
    """g <- g * (f - F^-1/2 eps): gradient of performer_features_ with respect to its `dash` argument."""
    _chk(g, f)
    assert g.shape == f.shape and g.is_contiguous() and f.is_contiguous() and g.numel() % 4 == 0
    call("srhip_performer_features_bwd", _p(g), _p(f), g.numel(), float(f.shape[1]) ** -0.5 * float(eps), _st())
    return g


def enlca_finish_bwd(dout, num, dnum, res_scale):
    _chk(dout, num, dnum)
    T, Cy = dout.shape
    assert dout.is_contiguous() and num.shape == dnum.shape and num.is_contiguous() and dnum.is_contiguous() and num.shape[1] > Cy
    call("srhip_enlca_finish_bwd", _p(dout), _p(num), num.stride(0), _p(dnum), T, Cy, float(res_scale), _st())
    return dnum


def performer_features_(dash, data, eps=1e-4):
    """dash[t][j] <- F^-1/2 (exp(dash[t][j] - |data[t]|^2 / 2) + eps), F = dash.shape[1] (softmax_kernel,
    network_enlcn.py:207-240)."""
    _chk(dash, data)
    assert dash.dim() == 2 and data.dim() == 2 and dash.shape[0] == data.shape[0] and dash.stride(1) == data.stride(1) == 1
    call("srhip_performer_features", _p(dash), dash.stride(0), _p(data), data.stride(0), dash.shape[0], dash.shape[1],
         data.shape[1], float(dash.shape[1]) ** -0.5, float(eps), _st())
    return dash


def enlca_finish(num, x, out, res_scale):
    """out = x + res_scale * num[:, :Cy] / num[:, Cy]  (x, out contiguous [T, Cy]; num [T, > Cy])."""
    _chk(num, x, out)
    T, Cy = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape and num.shape[0] == T and num.shape[1] > Cy
    call("srhip_enlca_finish", _p(num), num.stride(0), _p(x), _p(out), T, Cy, float(res_scale), _st())
    return out


def nlsa_hash_buckets(L, chunk_size):
    return lib.srhip_nlsa_hash_buckets(int(L), int(chunk_size))


def nlsa_order(x_embed, rotations, N, L):
    """Token order of every (sample, hash round) of NLSN's sparse attention (network_nlsn.py:145-207): x_embed [N*L, Ce],
    rotations [1, Ce, n_hashes, hash_buckets // 2] (what the reference draws with torch.randn at every call) -> int64
    [N, n_hashes, L], low 20 bits = token index, ordered by hash code then token."""
    _chk(x_embed, rotations)
    _, Ce, nh, hbh = rotations.shape
    rot_t = rotations[0].permute(1, 2, 0).reshape(nh * hbh, Ce).contiguous()        # [round-major column, Ce]
    rotated = gemm_nt(x_embed, rot_t)                                               # exact-f32 MFMA
    items = N * nh * L
    keys = torch.empty(items, dtype=torch.int64, device=x_embed.device)
    order = torch.empty(N, nh, L, dtype=torch.int64, device=x_embed.device)
    wsb = lib.srhip_nlsa_sort_ws(items)
    ws = SCRATCH.get("nlsa_sort", (wsb + 3) // 4, device=x_embed.device)
    call("srhip_nlsa_order", _p(rotated), rotated.stride(0), keys.data_ptr(), order.data_ptr(), _p(ws), wsb, N, L, nh, 2 * hbh,
         _st())
    return order


def nlsa_attention(x_embed, y_embed, order, x, out, N, L, chunk_size, res_scale, keep=False):
    """out = x + res_scale * NonLocalSparseAttention(x) given the embeddings and the token order (network_nlsn.py:209-266).
    keep (training): the per-round results ret [N, nh, L, Cy] and bucket scores [N, nh, L] (token positions) are fresh
    tensors, returned for the backward; else they live in the shared scratch."""
    _chk(x_embed, y_embed, x, out)
    Ce, Cy, nh = x_embed.shape[1], y_embed.shape[1], order.shape[1]
    assert x_embed.is_contiguous() and y_embed.is_contiguous() and x.is_contiguous() and out.is_contiguous()
    assert x.shape == (N * L, Cy) == tuple(out.shape)
    if keep:
        ret = torch.empty(N * nh * L * Cy, device=x.device)
        score = torch.empty(N * nh * L, device=x.device)
    else:
        ret = SCRATCH.get("nlsa_ret", N * nh * L * Cy, device=x.device)
        score = SCRATCH.get("nlsa_score", N * nh * L, device=x.device)
    args = (_p(x_embed), _p(y_embed), order.data_ptr(), _p(ret), _p(score), _p(x), _p(out), N, L, Ce, Cy,
            nh, int(chunk_size), float(res_scale), _st())
    if probe.on("nlsa"):
        Lp = -(-L // chunk_size) * chunk_size       # every (padded) token against its own and the two neighbouring chunks
        with probe.timed(("nlsa", N, L, Ce, Cy, nh), 2.0 * N * nh * Lp * 3 * chunk_size * (Ce + Cy),
                         4.0 * N * L * (nh * (Ce + Cy) * 4 + nh * (Cy + 1) * 2 + 2 * Cy)):
            call("srhip_nlsa_attention", *args)
    else:
        call("srhip_nlsa_attention", *args)
    if keep:
        return out, ret[:N * nh * L * Cy].view(N, nh, L, Cy), score[:N * nh * L].view(N, nh, L)
    return out


def unary(x, out, kind):
    """kind 'gelu' (nn.GELU(), exact erf) or 'sigmoid'; out may be x."""
    _chk(x, out)
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == out.numel()
    call("srhip_unary", _p(x), _p(out), x.numel(), {"gelu": 0, "sigmoid": 1}[kind], _st())
    return out


def unary_bwd(xy, g, out, kind):
    """out = g * f'(.): kind 'gelu' takes the op's input, 'sigmoid' its output."""
    _chk(xy, g, out)
    assert xy.is_contiguous() and g.is_contiguous() and out.is_contiguous() and xy.numel() == g.numel() == out.numel()
    call("srhip_unary_bwd", _p(xy), _p(g), _p(out), xy.numel(), {"gelu": 0, "sigmoid": 1}[kind], _st())
    return out


def fft2_mag_pow_shift(x, out, gamma=0.8, eps=1e-8):
    """out = fftshift2d((|fftn(x, dim=(H, W))| + eps) ** gamma) on NHWC [B, H, W, C] (network_dfcan.py:27-36,60-64)."""
    _chk(x, out)
    B, H, W, C = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
    ws = SCRATCH.get("fft2_ws", 2 * x.numel(), device=x.device)
    call("srhip_fft2_mag_pow_shift", _p(x), _p(out), _p(ws), B, H, W, C, float(gamma), float(eps), _st())
    return out


def fft2_mag_pow_shift_bwd(x, g, dx, gamma=0.8, eps=1e-8):
    """dx = gradient of fft2_mag_pow_shift at x for the output gradient g (NHWC, contiguous): the in-tree separable DFT, four
    passes (forward rows, forward columns with G . F formed in place, inverse columns, inverse rows' real part)."""
    _chk(x, g, dx)
    B, H, W, C = x.shape
    assert x.is_contiguous() and g.is_contiguous() and dx.is_contiguous() and g.shape == x.shape == dx.shape
    ws = SCRATCH.get("fft2_ws", 2 * x.numel(), device=x.device)
    call("srhip_fft2_mag_pow_shift_bwd", _p(x), _p(g), _p(dx), _p(ws), B, H, W, C, float(gamma), float(eps), _st())
    return dx


def channel_gate(feat, w1, b1, w2, b2, x0, x1, out, mid_act="relu"):
    """out = x0 + x1 * sigmoid(W2 relu(W1 mean_pixels(feat) + b1) + b2) per (sample, channel): the channel gate of DFCAN's
    RCAB (network_dfcan.py:65-70).  feat, x0, x1, out NHWC [B, H, W, C]; w1 [Cm, C], w2 [C, Cm]."""
    _chk(feat, w1, b1, w2, b2, x0, x1, out)
    B, H, W, C = feat.shape
    Cm = w1.shape[0]
    assert all(t.is_contiguous() for t in (feat, w1, w2, x1, out) + (() if x0 is None else (x0,)))
    assert w1.shape == (Cm, C) and w2.shape == (C, Cm)
    ws = SCRATCH.get("gate_ws", lib.srhip_channel_gate_ws(B, H * W, C), torch.float64, feat.device)
    gate = SCRATCH.get("gate_vec", B * C, device=feat.device)
    call("srhip_channel_gate", _p(feat), _p(w1), _p(b1), _p(w2), _p(b2), _p(x0), _p(x1), _p(out), _p(gate), _p(ws), B, H * W, C,
         Cm, {"relu": 0, "silu": 1}[mid_act], _st())
    return out


def unfold(x, C, k, s, pad, out, ldx=None):
    """F.unfold(x, k, stride=s, padding=pad) on NHWC data: x [B, H, W, >= C] (a channel-slice view is fine) -> out [B * nT,
    >= C*k*k] (row pitch out.stride(0)), columns c*k*k + ky*k + kx."""
    _chk(x, out)
    B, H, W = x.shape[:3]
    assert x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.stride(0) == H * x.stride(1) and out.stride(1) == 1
    call("srhip_unfold", _p(x), x.stride(2), _p(out), out.stride(0), B, H, W, C, k, s, pad, _st())
    return out


def fold(tok, C, k, s, out):
    """F.fold(tok, (H, W), k, stride=s) into NHWC out [B, H, W, >= C] (a channel-slice view is fine); tok [B * nT, >= C*k*k]."""
    _chk(tok, out)
    B, H, W = out.shape[:3]
    assert out.stride(3) == 1 and out.stride(1) == W * out.stride(2) and out.stride(0) == H * out.stride(1) and tok.stride(1) == 1
    call("srhip_fold", _p(tok), tok.stride(0), _p(out), out.stride(2), B, H, W, C, k, s, _st())
    return out


def layernorm_rows(x, gamma, beta, out, eps=1e-5):
    """nn.LayerNorm with affine over the rows of a 2-D view of any width; out may be x."""
    _chk(x, gamma, beta, out)
    assert x.dim() == 2 and out.shape == x.shape and x.stride(1) == 1 and out.stride(1) == 1
    call("srhip_layernorm_rows", _p(x), x.stride(0), _p(out), out.stride(0), _p(gamma), _p(beta), x.shape[0], x.shape[1],
         float(eps), _st())
    return out


def layernorm_rows_bwd(dy, x, gamma, dx, dgamma, dbeta, eps=1e-5):
    """backward of layernorm_rows: dx [M, C], dgamma [C], dbeta [C] from dy / x (2-D views, rows of at most 2048 values)."""
    _chk(dy, x, gamma, dx, dgamma, dbeta)
    assert x.dim() == 2 and dy.shape == x.shape == dx.shape and x.stride(1) == 1 and dy.stride(1) == 1 and dx.stride(1) == 1
    M, C = x.shape
    ws = SCRATCH.get("ln_rows_bwd_ws", lib.srhip_layernorm_rows_bwd_ws(M, C), device=x.device)
    call("srhip_layernorm_rows_bwd", _p(dy), dy.stride(0), _p(x), x.stride(0), _p(gamma), _p(dx), dx.stride(0), _p(dgamma),
         _p(dbeta), _p(ws), M, C, float(eps), _st())
    return dx


def layernorm_rows_res(x, res, gamma, beta, out, eps=1e-5):
    """out = res + nn.LayerNorm(x) over the rows of 2-D views of at most 256 columns; out may be x or res."""
    _chk(x, res, gamma, beta, out)
    assert x.dim() == 2 and out.shape == x.shape == res.shape and x.stride(1) == 1 and out.stride(1) == 1 and res.stride(1) == 1
    call("srhip_layernorm_rows_res", _p(x), x.stride(0), _p(res), res.stride(0), _p(out), out.stride(0), _p(gamma), _p(beta),
         x.shape[0], x.shape[1], float(eps), _st())
    return out


def softmax_rows_lse_(x, lse, scale=1.0):
    """softmax_rows_ that also leaves lse[r] = log sum exp(scale * x[r])."""
    _chk(x, lse)
    assert x.dim() == 2 and x.stride(1) == 1 and lse.numel() == x.shape[0]
    call("srhip_softmax_rows_lse", _p(x), x.stride(0), x.shape[0], x.shape[1], float(scale), _p(lse), _st())
    return x


def softmax_rows_bwd_(P, dP, dlse=None):
    """dP <- P * (dP - rowsum(P * dP) + dlse[:, None]): gradient with respect to the logits (and through lse)."""
    _chk(P, dP, dlse)
    assert P.shape == dP.shape and P.dim() == 2 and P.stride(1) == 1 and dP.stride(1) == 1 and P.stride(0) == dP.stride(0)
    call("srhip_softmax_rows_bwd", _p(P), _p(dP), P.stride(0), P.shape[0], P.shape[1], _p(dlse), _st())
    return dP


def rowdot(a, b, out=None):
    _chk(a, b, out)
    assert a.shape == b.shape and a.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(a.shape[0], device=a.device)
    call("srhip_rowdot", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), a.shape[0], a.shape[1], _st())
    return out


def softmax_rows_(x, scale=1.0):
    """x[r] <- softmax(scale * x[r]) in place, 2-D view."""
    _chk(x)
    assert x.dim() == 2 and x.stride(1) == 1
    call("srhip_softmax_rows", _p(x), x.stride(0), x.shape[0], x.shape[1], float(scale), _st())
    return x


def dwconv3x3(x, w, bias, out):
    """nn.Conv2d(C, C, 3, padding=1, groups=C) on NHWC x / out (channel-slice views are fine); w [C, 1, 3, 3]."""
    _chk(x, w, bias, out)
    B, H, W, C = out.shape
    assert x.stride(3) == 1 and out.stride(3) == 1 and w.is_contiguous() and w.numel() == 9 * C
    if probe.on("omni_ops"):
        with probe.timed(("omni_ops", "dwconv3x3", B * H * W, C), 18.0 * B * H * W * C, 8.0 * B * H * W * C):
            call("srhip_dwconv3x3", _p(x), x.stride(2), _p(w), _p(bias), _p(out), out.stride(2), B, H, W, C, _st())
    else:
        call("srhip_dwconv3x3", _p(x), x.stride(2), _p(w), _p(bias), _p(out), out.stride(2), B, H, W, C, _st())
    return out


def group_attention(qkv, bias, out, n, heads, scale):
    """softmax(scale q k^T + bias) v per (n consecutive rows, head): qkv [G*n, 3C] -> out [G*n, C]."""
    _chk(qkv, bias, out)
    C = out.shape[1]
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape == (out.shape[0], 3 * C) and out.shape[0] % n == 0
    if probe.on("omni_ops"):
        T = out.shape[0]
        with probe.timed(("omni_ops", "group_attention", T, C, n), 4.0 * T * n * C, 16.0 * T * C):
            call("srhip_group_attention", _p(qkv), _p(bias), _p(out), out.shape[0] // n, n, C, heads, float(scale), _st())
    else:
        call("srhip_group_attention", _p(qkv), _p(bias), _p(out), out.shape[0] // n, n, C, heads, float(scale), _st())
    return out


def channel_attention(qkv, temperature, out, heads, ps, grid):
    _chk(qkv, temperature, out)
    B, H, W, C = out.shape
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape == (B, H, W, 3 * C)
    if probe.on("omni_ops"):
        T = B * H * W
        with probe.timed(("omni_ops", "channel_attention", T, C, ps), 4.0 * T * C * (C // heads), 16.0 * T * C):
            call("srhip_channel_attention", _p(qkv), _p(temperature), _p(out), B, H, W, C, heads, ps, int(bool(grid)), _st())
    else:
        call("srhip_channel_attention", _p(qkv), _p(temperature), _p(out), B, H, W, C, heads, ps, int(bool(grid)), _st())
    return out


def gelu_gate(x, out):
    _chk(x, out)
    T, C = out.shape
    assert x.is_contiguous() and out.is_contiguous() and x.shape == (T, 2 * C)
    call("srhip_gelu_gate", _p(x), _p(out), T, C, _st())
    return out


def maxpool2d(x, k, s):
    _chk(x)
    B, H, W, C = x.shape
    out = torch.empty(B, (H - k) // s + 1, (W - k) // s + 1, C, device=x.device)
    call("srhip_maxpool2d", _p(x.contiguous()), _p(out), B, H, W, C, k, s, _st())
    return out


def bilinear_resize(x, Ho, Wo):
    _chk(x)
    B, H, W, C = x.shape
    out = torch.empty(B, Ho, Wo, C, device=x.device)
    call("srhip_bilinear_resize", _p(x.contiguous()), _p(out), B, H, W, C, Ho, Wo, _st())
    return out


def mul_sigmoid(x, g, out):
    _chk(x, g, out)
    assert x.is_contiguous() and g.is_contiguous() and out.is_contiguous() and x.numel() == g.numel() == out.numel()
    call("srhip_mul_sigmoid", _p(x), _p(g), _p(out), x.numel(), _st())
    return out


def mul(a, b, out=None):
    """out = a * b element-wise (same shapes, dense)."""
    _chk(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and a.numel() == b.numel() == out.numel()
    call("srhip_mul", _p(a), _p(b), _p(out), a.numel(), _st())
    return out


def add_periodic(x, v):
    """x.view(-1)[i] += v.view(-1)[i % v.numel()] (x dense, its size a multiple of v's)."""
    _chk(x, v)
    assert x.is_contiguous() and v.is_contiguous()
    call("srhip_add_periodic", _p(x), _p(v), x.numel(), v.numel(), _st())
    return x


def sum_periodic(x, out):
    """out.view(-1)[j] = sum_k x.view(-1)[k * out.numel() + j]."""
    _chk(x, out)
    assert x.is_contiguous() and out.is_contiguous()
    call("srhip_sum_periodic", _p(x), _p(out), x.numel(), out.numel(), _st())
    return out


def maxpool2d_bwd(x, g, k, s):
    _chk(x, g)
    B, H, W, C = x.shape
    assert x.is_contiguous() and g.is_contiguous() and tuple(g.shape) == (B, (H - k) // s + 1, (W - k) // s + 1, C)
    dx = torch.empty_like(x)
    call("srhip_maxpool2d_bwd", _p(x), _p(g), _p(dx), B, H, W, C, k, s, _st())
    return dx


def avgpool2d(x, k):
    """nn.AvgPool2d(k, k) on NHWC [B, H, W, C] (AnchorLinear, network_grl.py:603-620)."""
    _chk(x)
    B, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty(B, H // k, W // k, C, device=x.device)
    call("srhip_avgpool2d", _p(x), _p(out), B, H, W, C, k, _st())
    return out


def cpb_bias(table, index, heads):
    """16 sigmoid(table[index]) as the key-major bias image [heads, N2, N1] of srhip_cosine_window_attention; table
    [entries, heads] f32 (the CPB MLP's output), index [N1, N2] int64 (network_grl.py:305-311)."""
    _chk(table)
    N1, N2 = index.shape
    assert table.is_contiguous() and table.shape[1] == heads and index.is_contiguous() and index.dtype == torch.int64
    assert index.device == table.device
    out = torch.empty(heads, N2, N1, device=table.device)
    call("srhip_cpb_bias", _p(table), _p(index), _p(out), heads, N1, N2, table.shape[0], _st())
    return out


def cosine_window_attention(q, qwin, k, v, kwin, logit_scale, biasT, out, heads, d, shift=0):
    """GRL's Attention.attn (network_grl.py:338-355) between the windows (qwin = (wh, ww)) of the NHWC image q [B, qH, qW, >=
    heads*d] and the windows kwin of the NHWC images k, v [B, kH, kW, .] on the same window grid; q / k / v / out may be
    channel-slice views of wider images (stride(3) == 1).  The result is written at the query tokens' own pixels of out."""
    for t in (q, k, v, out):
        assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 4 and t.stride(3) == 1
        assert t.stride(1) == t.shape[2] * t.stride(2) and t.stride(0) == t.shape[1] * t.stride(1)
    _chk(logit_scale, biasT)
    B, qH, qW = q.shape[:3]
    kH, kW = k.shape[1:3]
    assert v.shape[:3] == k.shape[:3] and out.shape[:3] == q.shape[:3] and k.shape[0] == B
    assert biasT.shape == (heads, kwin[0] * kwin[1], qwin[0] * qwin[1]) and biasT.is_contiguous()
    assert logit_scale.numel() == heads and logit_scale.is_contiguous()
    args = (q.data_ptr(), q.stride(2), qH, qW, qwin[0], qwin[1], k.data_ptr(), k.stride(2),
            v.data_ptr(), v.stride(2), kH, kW, kwin[0], kwin[1], _p(logit_scale), _p(biasT), out.data_ptr(), out.stride(2), B,
            heads, d, int(shift), _st())
    if probe.on("grl_attn"):
        nq, nk = qwin[0] * qwin[1], kwin[0] * kwin[1]
        with probe.timed(("grl_attn", qH, qW, nq, nk, heads, d), 4.0 * B * qH * qW * nk * heads * d,
                         4.0 * B * heads * d * (2 * qH * qW + 2 * kH * kW)):
            call("srhip_cosine_window_attention", *args)
    else:
        call("srhip_cosine_window_attention", *args)
    return out


def mlp_f16_fusable(C, hidden):
    """Shapes srhip_mlp_fwd_f16x2 / srhip_mlp_bwd_f16x2 take with the weight planes PrepTable.linear builds for the
    Linear GEMMs (format 1: two fp16 planes)."""
    def fmt1(rows, kd):
        return F16X2 and (rows % 180 == 0 or (rows > 128 and rows % 128 != 0)) and kd <= 1024
    return (use_bx3() and C % 4 == 0 and 4 <= C <= 192 and hidden % 4 == 0 and 4 <= hidden <= 384
            and fmt1(hidden, C) and fmt1(C, hidden))


def mlp_fwd_f16(x, stats, W1, b1, W2, b2, out, h=None, rowscale=None, rows_per_scale=1, stats_out=None):
    """out = x + s * (gelu(LN(x) . W1^T + b1) . W2^T + b2) in ONE kernel on the Linear GEMMs' own operands
    (W1 = planes of W1*gamma [hidden, C], W2 = planes of W2 [C, hidden], both format 1); h (optional) receives the
    pre-activation, stats_out the {mean, rstd} of the out rows."""
    _chk(x, stats, b1, b2, out, h, rowscale, stats_out)
    M, C = x.shape
    hidden = b1.shape[0]
    assert W1.fmt == 1 and W2.fmt == 1 and (W1.rows, W1.K) == (hidden, C) and (W2.rows, W2.K) == (C, hidden)
    assert out.shape == (M, C) and (h is None or h.shape == (M, hidden))
    args = (_p(x), x.stride(0), _p(stats), _p(W1.planes), _p(b1), _p(W2.planes), _p(b2), _p(h),
            0 if h is None else h.stride(0), _p(out), out.stride(0), M, C, hidden, _p(rowscale), rows_per_scale,
            _p(stats_out), _st())
    if probe.on("mlp_fused"):
        with probe.timed(("mlp_fused", M, C, hidden, "fwd"), 4.0 * M * C * hidden,
                         4.0 * (M * (2 * C + (hidden if h is not None else 0)) + 2 * C * hidden),
                         kernel="k_mlp_f16<false", lb_bytes=4.0 * (M * 2 * C + 2 * C * hidden)):
            call("srhip_mlp_fwd_f16x2", *args)
    else:
        call("srhip_mlp_fwd_f16x2", *args)
    return out


def wmsa_f16_fusable(C, heads):
    """Shapes srhip_wmsa_fwd_f16x2 takes with the planes PrepTable.linear builds (format 1) and the bias images of
    the fp16x2 attention."""
    def fmt1(rows, kd):
        return F16X2 and (rows % 180 == 0 or (rows > 128 and rows % 128 != 0)) and kd <= 1024
    return (wattn_f16_ok(C, heads) and C % 4 == 0 and 4 <= C <= 192 and heads <= 8 and fmt1(3 * C, C) and fmt1(C, C))


def wmsa_fwd_f16(x, stats, Wq, bq, Wp, bp, biasF, qkv, att, out, B, H, W, heads, shift, rowscale=None,
                 stats_out=None):
    """The W-MSA half of a Swin block, forward, in ONE kernel: qkv = LN(x) . Wq^T + bq, att = window attention,
    out = x + s * (att . Wp^T + bp).  Wq = planes of Wqkv*gamma [3C, C], Wp = planes of Wproj [C, C] (format 1)."""
    _chk(x, stats, bq, bp, biasF, qkv, att, out, rowscale, stats_out)
    T, C = x.shape
    assert T == B * H * W and Wq.fmt == 1 and Wp.fmt == 1 and (Wq.rows, Wq.K) == (3 * C, C) and (Wp.rows, Wp.K) == (C, C)
    assert (qkv is None and heads in (5, 6)) or qkv.shape == (T, 3 * C)      # None (inference): q, k, v stay in registers
    assert att.shape == (T, C) and out.shape == (T, C)
    assert rowscale is None or rowscale.numel() == B
    args = (_p(x), _p(stats), _p(Wq.planes), _p(bq), _p(Wp.planes), _p(bp), _p(biasF), _p(rowscale), _p(qkv),
            _p(att), _p(out), _p(stats_out), B, H, W, C, heads, shift, _st())
    if probe.on("wmsa_fused"):
        with probe.timed(("wmsa_fused", T, C, heads, "fwd"), 2.0 * T * C * 4 * C + 4.0 * T * 64 * C,
                         4.0 * (T * ((6 if qkv is not None else 3) * C) + 4 * C * C),
                         kernel="k_wmsa_f16h", lb_bytes=4.0 * (T * 2 * C + 4 * C * C)):
            call("srhip_wmsa_fwd_f16x2", *args)
    else:
        call("srhip_wmsa_fwd_f16x2", *args)
    return out


def mlp_bwd_f16(dy, W2T, W1T, h, dh, gh, x, stats, dx, rowscale=None, rows_per_scale=1, chain=None, front=None):
    """Data gradient of mlp_fwd_f16 in ONE kernel: dh = (s * dy . W2) * gelu'(h), gh = gelu(h), dx = dy +
    LayerNorm_backward(dh . W1f; x, stats).  W2T = planes of W2^T [hidden, C], W1T = planes of (W1*gamma)^T [C, hidden].
    chain = (W3, out3, rowscale3): out3 = s3 * (dx . W3^T) behind it in the same kernel (W3 = planes [C, C], format 1:
    the data gradient of the attention's proj Linear).
    front = (X0, W0, x0, stats0, res0): dy is COMPUTED in the kernel (and written): dy = res0 + LayerNorm_backward(X0 @
    W0^T; x0, stats0) -- the qkv Linear's data gradient of the Swin block behind this one (W0 = planes [C, K0])."""
    _chk(dy, h, dh, gh, x, stats, dx, rowscale)
    M, C = dy.shape
    hidden = h.shape[1]
    assert W2T.fmt == 1 and W1T.fmt == 1 and (W2T.rows, W2T.K) == (hidden, C) and (W1T.rows, W1T.K) == (C, hidden)
    # gh None: gelu(h) is not stored -- the fc2 weight gradient reads h with b_mode=2 (linear_wgrad_grouped)
    assert dh.shape == h.shape and dh.stride(0) == h.stride(0)
    assert gh is None or (gh.shape == h.shape and gh.stride(0) == h.stride(0))
    assert x.shape == (M, C) and dx.shape == (M, C)
    args = (_p(dy), dy.stride(0), _p(W2T.planes), _p(W1T.planes), _p(h), h.stride(0), _p(dh), _p(gh), _p(x),
            x.stride(0), _p(stats), _p(dx), dx.stride(0), M, C, hidden, _p(rowscale), rows_per_scale, _st())
    name = "srhip_mlp_bwd_f16x2"
    if chain is not None:
        W3, out3, rs3 = chain
        _chk(out3, rs3)
        assert W3.fmt == 1 and (W3.rows, W3.K) == (C, C) and out3.shape == (M, C)
        args = args[:-1] + (_p(W3.planes), _p(out3), out3.stride(0), _p(rs3), _st())
        name = "srhip_mlp_bwd_chain_f16x2"
    if front is not None:
        X0, W0, x0, st0, res0 = front
        _chk(X0, x0, st0, res0)
        K0 = X0.shape[1]
        assert W0.fmt == 1 and (W0.rows, W0.K) == (C, K0) and X0.shape[0] == M and x0.shape == (M, C) == res0.shape
        if chain is None:
            args = args[:-1] + (None, None, 0, None, _st())
        args = (_p(X0), X0.stride(0), K0, _p(W0.planes), _p(x0), x0.stride(0), _p(st0), _p(res0), res0.stride(0)) + args
        name = "srhip_mlp_bwd_front_chain_f16x2"
    if probe.on("mlp_fused"):
        tag = ("qkv+" if front is not None else "") + "bwd" + ("+proj" if chain is not None else "")
        k0 = front[0].shape[1] if front is not None else 0
        with probe.timed(("mlp_fused", M, C, hidden, tag),
                         4.0 * M * C * hidden + (0 if chain is None else 2.0 * M * C * C) + 2.0 * M * C * k0,
                         4.0 * (M * (3 * C + (3 if gh is not None else 2) * hidden + (C if chain is not None else 0)
                                     + (k0 + 2 * C if front else 0)) + 2 * C * hidden),
                         kernel="k_mlp_f16<true",
                         # fused bound: incoming / outgoing gradients, the block's saved input rows, weights -- not h / dh / gelu(h)
                         lb_bytes=4.0 * (M * (3 * C + (C if chain is not None else 0) + (k0 + 2 * C if front else 0))
                                         + 2 * C * hidden + (C * C if chain is not None else 0) + C * k0)):
            call(name, *args)
    else:
        call(name, *args)
    return dx


def conv3x3(X, Wp, bias, Cout, out=None, epi=0, R=None, rowscale=None, alpha=1.0, in_bn=None, slope=None):
    """X NHWC [B,H,W,Cin], Wp packed [9,Cout,Cin] (f32 tensor or Bx3) -> [B,H,W,Cout].
    in_bn (coef [4, Cin] of srhip_bn_apply; fp16x2 weight planes) / slope (PReLU's one-element parameter) / epi 8-10: the
    folds of srhip_conv3x3_nhwc_split_ex."""
    bx = isinstance(Wp, Bx3)
    _chk(X, None if bx else Wp, bias, out, R, rowscale, in_bn, slope)
    B, H, W, Cin = X.shape
    if out is None:
        out = torch.empty(B, H, W, Cout, device=X.device, dtype=torch.float32)
    if bx:
        assert Wp.rows == 9 * Cout and Wp.K == Cin
    args = (_p(X), X.stride(2), _p(Wp.planes if bx else Wp), _p(bias), _p(out), out.stride(2),
            B, H, W, Cin, Cout, epi, _p(R), 0 if R is None else R.stride(2), _p(rowscale),
            float(alpha), _st())
    name = ("srhip_conv3x3_nhwc_f16x2" if Wp.fmt == 1 else "srhip_conv3x3_nhwc_bx3") if bx else "srhip_conv3x3_nhwc"
    if in_bn is not None or slope is not None or epi >= 8:     # epi 8-11
        assert bx, "conv3x3: prologue / epilogues 8-10 run on the split-operand kernels"
        assert in_bn is None or (Wp.fmt == 1 and tuple(in_bn.shape) == (4, Cin) and in_bn.is_contiguous())
        name = "srhip_conv3x3_nhwc_split_ex"
        args = (int(Wp.fmt),) + args[:-1] + (_p(in_bn), _p(slope), _st())
    kind = "conv_nt" if bx else "conv_nt_f32"
    if probe.on(kind):
        T = B * H * W
        with probe.timed((kind, T, Cout, Cin), 18.0 * T * Cout * Cin,
                         4.0 * (T * Cin + 9 * Cin * Cout + T * Cout * (2 if R is not None else 1))):
            call(name, *args)
    else:
        call(name, *args)
    return out


def _ph(t):
    """pointer of an fp16 CUDA tensor (or None)"""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float16:
        raise SrhipError("fp16-storage ops take float16 CUDA/HIP tensors")
    return t.data_ptr()


def conv3x3_h16(X, Wp, bias, Cout, out=None, epi=0, R=None, alpha=1.0, ps2=False, in_bn=None, center_only=False):
    """fp16-storage 3x3 conv (srhip_conv3x3_nhwc_h16): X NHWC float16 [B,H,W,Cin], Wp = the fp16x2 weight planes (Bx3 fmt 1;
    ps2: built with PrepTable.conv(ps2=True)) -> float16 [B,H,W,Cout] (ps2: [B,2H,2W,Cout/4])."""
    assert isinstance(Wp, Bx3) and Wp.fmt == 1, "conv3x3_h16: fp16x2 weight planes"
    B, H, W, Cin = X.shape
    assert Wp.rows == 9 * Cout and Wp.K == Cin and X.stride(3) == 1 and X.stride(1) == W * X.stride(2)
    if out is None:
        out = torch.empty((B, 2 * H, 2 * W, Cout // 4) if ps2 else (B, H, W, Cout), device=X.device, dtype=torch.float16)
    _chk(bias, in_bn)
    assert in_bn is None or (tuple(in_bn.shape) == (4, Cin) and in_bn.is_contiguous())
    def run():
        call("srhip_conv3x3_nhwc_h16", _ph(X), X.stride(2), _p(Wp.planes), _p(bias), _ph(out), out.stride(2), B, H, W, Cin, Cout,
             int(epi), _ph(R), 0 if R is None else R.stride(2), float(alpha), int(bool(ps2)), _p(in_bn), int(bool(center_only)), _st())
    if probe.on("conv_h16"):
        T = B * H * W
        taps = 1 if center_only else 9
        with probe.timed(("conv_h16", T, Cout, Cin, taps), 2.0 * taps * T * Cout * Cin,
                         2.0 * (T * Cin + T * Cout + (T * Cout if R is not None else 0)) + 2.0 * taps * Cin * Cout):
            run()
    else:
        run()
    return out


def conv3x3_cin1_h16(x, w, bias, Co, out=None, relu=False, leaky=None):
    """f32 image [B,H,W] -> float16 features [B,H,W,Co] (w [Co,1,3,3] f32); relu / leaky = slope: the activation behind it."""
    _chk(x, w, bias)
    B, H, W = x.shape
    assert x.is_contiguous() and w.is_contiguous() and tuple(w.shape) == (Co, 1, 3, 3)
    if out is None:
        out = torch.empty(B, H, W, Co, device=x.device, dtype=torch.float16)
    call("srhip_conv3x3_cin1_h16", _p(x), _p(w), _p(bias), _ph(out), out.stride(2), B, H, W, Co,
         2 if leaky is not None else int(bool(relu)), float(leaky or 0.0), _st())
    return out


def conv3x3_cout1_h16(x, w, bias, add=None, out=None, in_bn=None):
    """float16 features [B,H,W,Ci] (through BatchNorm-ReLU if in_bn [4, Ci] is given) -> f32 image [B,H,W] (w [1,Ci,3,3] f32),
    + bias + the f32 image `add`."""
    _chk(w, bias, add, out, in_bn)
    B, H, W, Ci = x.shape
    assert x.stride(3) == 1 and w.is_contiguous() and tuple(w.shape) == (1, Ci, 3, 3) and (add is None or add.is_contiguous())
    if out is None:
        out = torch.empty(B, H, W, device=x.device, dtype=torch.float32)
    call("srhip_conv3x3_cout1_h16", _ph(x), x.stride(2), _p(w), _p(bias), _p(add), _p(in_bn), _p(out), B, H, W, Ci, _st())
    return out


def srcnn_fwd_h16(patches, W1p, b1, W2p, b2, w3, b3, out, image=None):
    """SRCNN's three layers in one launch (srhip_srcnn_fwd_h16): patches [T, 32] float16 (or None with image [B, H, W] f32: the
    patch matrix is built inside), W1p / W2p = centre-tap fp16x2 conv operands of [1024, 32] / [128, 1024], w3 [128], b3 [1]
    -> out [T] f32."""
    _chk(b1, b2, w3, b3, out, image)
    assert W1p.fmt == 1 and W2p.fmt == 1 and (W1p.rows, W1p.K) == (9 * 1024, 32) and (W2p.rows, W2p.K) == (9 * 128, 1024)
    if image is not None:
        assert patches is None and image.is_contiguous() and image.dim() == 3
        B, H, W = image.shape
        T = B * H * W
    else:
        T = patches.shape[0]
        B = H = W = 0
        assert patches.is_contiguous() and tuple(patches.shape) == (T, 32)
    assert w3.numel() == 128 and out.numel() == T
    def run():
        call("srhip_srcnn_fwd_h16", _ph(patches), _p(image), B, H, W, _p(W1p.planes), _p(b1), _p(W2p.planes), _p(b2), _p(w3), _p(b3),
             _p(out), T, _st())
    if probe.on("conv_h16"):        # 25 -> 1024 -> 128 -> 1 per pixel; the image in and out (the hidden maps never exist)
        with probe.timed(("conv_h16", "srcnn", T), 2.0 * T * (32 * 1024 + 1024 * 128 + 128), 8.0 * T + 2.0 * (32 * 1024 + 1024 * 128)):
            run()
    else:
        run()
    return out


def tn_plan(M, NI, NJ, conv=False, bx=None):
    S = ctypes.c_int(0)
    n = ctypes.c_long(0)
    call("srhip_tn_plan" + _tn_sfx(bx), M, NI, NJ, int(conv), ctypes.addressof(S), ctypes.addressof(n))
    return S.value, n.value


# Generation of the persistent buffers: bumped whenever a named buffer is REPLACED (a scratch buffer grown, an engine buffer
# re-made for another shape) -- the old tensor goes back to torch's caching allocator while a captured hipGraph may still hold
# its address.  TrainStep.step_graph / ModelPlain's evaluation graphs record it at capture and re-capture when it has moved
# (a validation forward on a larger image between two replays grew 'fft2_ws' / 'nlsa_sort' under a captured step).
_REALLOC_GEN = [0]


def realloc_generation():
    return _REALLOC_GEN[0]


def note_realloc():
    _REALLOC_GEN[0] += 1


class Scratch:
    """Grow-only device scratch buffers keyed by name (caller-owned workspace of
    the C-ABI; reused across calls so the hot loop never allocates)."""

    def __init__(self):
        self.bufs = {}

    def get(self, name, n, dtype=torch.float32, device="cuda"):
        b = self.bufs.get(name)
        if b is None or b.numel() < n or b.dtype != dtype:
            if b is not None:
                note_realloc()
            b = torch.empty(max(int(n), 1), device=device, dtype=dtype)
            self.bufs[name] = b
        return b


SCRATCH = Scratch()


def linear_wgrad(dY, X, dW, db, a_rowscale=None, a_rowscale_rows=1, b_mode=0, ln_stats=None,
                 ln=None):
    """dW[N,K] = dY[M,N]^T . pro(X)[M,K]; db = colsum(dY).  ``ln`` =
    (W, gamma, beta, dgamma, dbeta) finishes a LayerNorm-folded Linear."""
    _chk(dY, X, dW, db, a_rowscale, ln_stats)
    M, N = dY.shape
    K = X.shape[1]
    S, n = tn_plan(M, N, K)
    part = SCRATCH.get("tn_part", n, device=dY.device)
    cs = SCRATCH.get("tn_colsum", S * N, device=dY.device)
    call("srhip_gemm_tn" + _tn_sfx(), _p(dY), dY.stride(0), _p(X), X.stride(0), M, N, K, _p(a_rowscale),
         a_rowscale_rows, b_mode, _p(ln_stats), _p(part), _p(cs), S, _st())
    if ln is None:
        call("srhip_reduce_linear_wgrad", _p(part), _p(cs), S, _p(dW), _p(db), N, K, _st())
    else:
        W, gamma, beta, dgamma, dbeta = ln
        lnws = SCRATCH.get("ln_affine_ws", lib.srhip_ln_affine_ws(N, K), device=dY.device)
        call("srhip_reduce_ln_linear_wgrad", _p(part), _p(cs), S, _p(W), _p(gamma), _p(beta),
             _p(dW), _p(db), _p(dgamma), _p(dbeta), N, K, _p(lnws), _st())


class _TnProblem(ctypes.Structure):   # srhip_tn_problem (include/srhip.h)
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_long), ("B", ctypes.c_void_p),
                ("ldb", ctypes.c_long), ("NI", ctypes.c_int), ("NJ", ctypes.c_int),
                ("a_rowscale", ctypes.c_void_p), ("a_rowscale_rows", ctypes.c_int),
                ("b_mode", ctypes.c_int), ("ln_stats", ctypes.c_void_p),
                ("part", ctypes.c_void_p), ("part_colsum", ctypes.c_void_p)]


class _ReduceProblem(ctypes.Structure):   # srhip_reduce_problem (include/srhip.h)
    _fields_ = [("part", ctypes.c_void_p), ("colsum", ctypes.c_void_p), ("W", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("dW", ctypes.c_void_p),
                ("db", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("ln_ws", ctypes.c_void_p)]


def linear_wgrad_grouped(problems):
    """Up to 4 (exact-f32 kernels) / 24 (bf16x3) Linear weight-gradient problems over the same rows in ONE launch.
    Each problem: dict(dY, X, dW, db, a_rowscale=None, a_rowscale_rows=1, b_mode=0,
    ln_stats=None, ln=None) with the meaning of linear_wgrad()."""
    n = len(problems)
    M = problems[0]["dY"].shape[0]
    dev = problems[0]["dY"].device
    tiles = sum(lib.srhip_tn_tiles(q["dY"].shape[1], q["X"].shape[1]) for q in problems)
    S = ctypes.c_int(0)
    call("srhip_tn_group_plan" + _tn_sfx(), M, tiles, ctypes.addressof(S))
    S = S.value
    sizes = [(q["dY"].shape[1] * q["X"].shape[1], q["dY"].shape[1]) for q in problems]
    part = SCRATCH.get("tng_part", S * sum(a for a, _ in sizes), device=dev)
    cs = SCRATCH.get("tng_colsum", S * sum(b for _, b in sizes), device=dev)
    arr = (_TnProblem * n)()
    views = []
    po = co = 0
    for k, q in enumerate(problems):
        dY, X = q["dY"], q["X"]
        _chk(dY, X, q["dW"], q["db"], q.get("a_rowscale"), q.get("ln_stats"))
        N, K = dY.shape[1], X.shape[1]
        pk, ck = part[po:po + S * N * K], cs[co:co + S * N]
        po += S * N * K
        co += S * N
        views.append((pk, ck))
        a = arr[k]
        a.A, a.lda, a.B, a.ldb, a.NI, a.NJ = _p(dY), dY.stride(0), _p(X), X.stride(0), N, K
        a.a_rowscale, a.a_rowscale_rows = _p(q.get("a_rowscale")), q.get("a_rowscale_rows", 1)
        a.b_mode, a.ln_stats = q.get("b_mode", 0), _p(q.get("ln_stats"))
        a.part, a.part_colsum = _p(pk), _p(ck)
    if probe.on("linear_tn"):
        fl = sum(2.0 * M * q["dY"].shape[1] * q["X"].shape[1] for q in problems)
        by = sum(4.0 * (M * (q["dY"].shape[1] + q["X"].shape[1]) + q["dY"].shape[1] * q["X"].shape[1]) for q in problems)
        with probe.timed(("linear_tn", M, n), fl, by):
            call("srhip_gemm_tn_grouped" + _tn_sfx(), ctypes.addressof(arr), n, M, S, _st())
    else:
        call("srhip_gemm_tn_grouped" + _tn_sfx(), ctypes.addressof(arr), n, M, S, _st())
    red = (_ReduceProblem * n)()
    # LayerNorm-folded problems: one workspace slice each for the row blocks' dgamma / dbeta shares (deterministic
    # two-stage sum; no atomics, nothing to zero)
    lnsz = [lib.srhip_ln_affine_ws(q["dY"].shape[1], q["X"].shape[1]) if q.get("ln") is not None else 0 for q in problems]
    lnws = SCRATCH.get("ln_affine_ws_g", max(1, sum(lnsz)), device=dev)
    lo = 0
    for r, q, (pk, ck), lsz in zip(red, problems, views, lnsz):
        r.part, r.colsum, r.dW, r.db = _p(pk), _p(ck), _p(q["dW"]), _p(q["db"])
        r.N, r.K = q["dY"].shape[1], q["X"].shape[1]
        if q.get("ln") is not None:
            W, gamma, beta, dgamma, dbeta = q["ln"]
            r.W, r.gamma, r.beta, r.dgamma, r.dbeta = _p(W), _p(gamma), _p(beta), _p(dgamma), _p(dbeta)
            r.ln_ws = _p(lnws[lo:lo + lsz])
            lo += lsz
    call("srhip_reduce_wgrad_grouped", ctypes.addressof(red), n, S, _st())


def ps2_fusable(Cin, Cout):
    """conv Cin -> Cout followed by PixelShuffle(2) as one kernel per direction (conv3x3_ps2*)?"""
    return use_bx3() and Cout % 256 == 0 and Cin % 4 == 0 and Cin <= 128 and bx3_for(Cout, Cin) and bx3_nt_for(Cout, Cin)


def resblock64_fusable(F, Wp=None):
    """The one-launch ResBlock kernels (resblock.hip) take 64 -> 64 -> 64 channels on fp16x2 conv operands."""
    return F == 64 and F16X2_CONV and (Wp is None or (isinstance(Wp, Bx3) and Wp.fmt == 1))


def resblock64_fwd(x, W1, b1, W2, b2, res_scale, a, out):
    """a = relu(conv3x3(x; W1) + b1), out = x + res_scale * (conv3x3(a; W2) + b2) in ONE kernel (ResBlock.forward,
    network_nlsn.py:72-93).  x, a, out NHWC [B,H,W,64]; W1 / W2: fp16x2 conv operands (PrepTable.conv)."""
    _chk(x, b1, b2, a, out)
    B, H, W, C = x.shape
    assert C == 64 and a.shape == x.shape == out.shape and x.stride(3) == 1 and x.stride(1) == W * x.stride(2)
    assert all(isinstance(w, Bx3) and w.fmt == 1 and (w.rows, w.K) == (9 * 64, 64) for w in (W1, W2))
    def run():
        call("srhip_resblock64_fwd_f16x2", _p(x), x.stride(2), _p(W1.planes), _p(b1), _p(W2.planes), _p(b2), float(res_scale),
             _p(a), a.stride(2), _p(out), out.stride(2), B, H, W, _st())
    if probe.on("conv_nt"):
        T = B * H * W
        with probe.timed(("conv_nt", T, 64, 64, "resblock fwd"), 2 * 18.0 * T * 64 * 64, 4.0 * (3 * T * 64 + 2 * 9 * 64 * 64),
                         kernel="k_resblock64<false"):
            run()
    else:
        run()
    return out


def resblock64_bwd(g, W2T, W1T, a, res_scale, da, dx):
    """da = res_scale * conv3x3(g; W2^T) * (a > 0), dx = g + conv3x3(da; W1^T) in ONE kernel: the data gradient of
    resblock64_fwd.  W2T / W1T: the data-gradient conv operands (PrepTable.conv(data_grad=True))."""
    _chk(g, a, da, dx)
    B, H, W, C = g.shape
    assert C == 64 and a.shape == g.shape == da.shape == dx.shape and g.stride(3) == 1 and g.stride(1) == W * g.stride(2)
    assert all(isinstance(w, Bx3) and w.fmt == 1 and (w.rows, w.K) == (9 * 64, 64) for w in (W2T, W1T))
    def run():
        call("srhip_resblock64_bwd_f16x2", _p(g), g.stride(2), _p(W2T.planes), _p(W1T.planes), _p(a), a.stride(2),
             float(res_scale), _p(da), da.stride(2), _p(dx), dx.stride(2), B, H, W, _st())
    if probe.on("conv_nt"):
        T = B * H * W
        with probe.timed(("conv_nt", T, 64, 64, "resblock bwd"), 2 * 18.0 * T * 64 * 64, 4.0 * (4 * T * 64 + 2 * 9 * 64 * 64),
                         kernel="k_resblock64<true"):
            run()
    else:
        run()
    return dx


def conv3x3_ps2(X, Wp, bias, out, epi=0, alpha=1.0):
    """PixelShuffle(2)(conv3x3(X) + bias) in one kernel: X NHWC [B,H,W,Cin], Wp Bx3 pack built with
    PrepTable.conv(ps2=True) -> out NHWC [B,2H,2W,Cout/4]."""
    _chk(X, bias, out)
    B, H, W, Cin = X.shape
    F = out.shape[3]
    assert isinstance(Wp, Bx3) and Wp.rows == 9 * 4 * F and Wp.K == Cin and out.shape == (B, 2 * H, 2 * W, F)
    args = (_p(X), X.stride(2), _p(Wp.planes), _p(bias), _p(out), out.stride(2), B, H, W, Cin, 4 * F, epi,
            float(alpha), _st())
    fn = "srhip_conv3x3_ps2_f16x2" if Wp.fmt == 1 else "srhip_conv3x3_ps2_bx3"
    if probe.on("conv_nt"):
        T = B * H * W
        with probe.timed(("conv_nt", T, 4 * F, Cin, "ps2"), 18.0 * T * 4 * F * Cin, 4.0 * (T * Cin + 36 * F * Cin + T * 4 * F)):
            call(fn, *args)
    else:
        call(fn, *args)
    return out


def conv3x3_ps2_bwd_data(dYup, Wpt, out, epi=0, R=None, alpha=1.0):
    """Data gradient of conv3x3_ps2 read from the gradient of its OUTPUT: dYup NHWC [B,2H,2W,F], Wpt Bx3 pack
    built with PrepTable.conv(data_grad=True, ps2=True) -> out NHWC [B,H,W,Cin]; epi 4 / 7: the (Leaky)ReLU mask
    of the activation R that fed the conv."""
    _chk(dYup, out, R)
    B, H, W, Cin = out.shape
    F = dYup.shape[3]
    assert isinstance(Wpt, Bx3) and Wpt.rows == 9 * Cin and Wpt.K == 4 * F and dYup.shape == (B, 2 * H, 2 * W, F)
    args = (_p(dYup), dYup.stride(2), _p(Wpt.planes), _p(out), out.stride(2), B, H, W, 4 * F, Cin, epi, _p(R),
            0 if R is None else R.stride(2), float(alpha), _st())
    fn = "srhip_conv3x3_ps2_bwd_data_f16x2" if Wpt.fmt == 1 else "srhip_conv3x3_ps2_bwd_data_bx3"
    if probe.on("conv_nt"):
        T = B * H * W
        with probe.timed(("conv_nt", T, Cin, 4 * F, "ps2"), 18.0 * T * 4 * F * Cin, 4.0 * (T * Cin + 36 * F * Cin + T * 4 * F)):
            call(fn, *args)
    else:
        call(fn, *args)
    return out


def conv3x3_wgrad(dY, X, dW, db, ps2=False):
    """dY NHWC [B,H,W,Cout], X NHWC [B,H,W,Cin] -> dW [Cout,Cin,3,3], db [Cout].
    ps2: dY is the gradient of the PixelShuffle(2) OUTPUT, NHWC [B,2H,2W,Cout/4] (conv3x3_ps2)."""
    _chk(dY, X, dW, db)
    B, H, W, Cin = X.shape
    Cout = dW.shape[0]
    assert dY.shape == ((B, 2 * H, 2 * W, Cout // 4) if ps2 else (B, H, W, Cout))
    bx = bx3_for(Cout, Cin)
    assert bx or not ps2, "the fused PixelShuffle weight gradient runs on the bf16x3 kernels"
    S, n = tn_plan(B * H * W, Cout, Cin, True, bx)
    part = SCRATCH.get("tn_part", n, device=dY.device)
    cs = SCRATCH.get("tn_colsum", S * Cout, device=dY.device)
    def run():
        name = "srhip_conv3x3_ps2_wgrad_bx3" if ps2 else "srhip_conv3x3_wgrad" + _tn_sfx(bx)
        call(name, _p(dY), dY.stride(2), _p(X), X.stride(2), B, H, W, Cout, Cin,
             _p(part), _p(cs), S, _st())
    if probe.on("conv_tn"):
        T = B * H * W
        with probe.timed(("conv_tn", T, Cout, Cin), 18.0 * T * Cout * Cin, 4.0 * (T * Cin + T * Cout + 9 * Cin * Cout)):
            run()
    else:
        run()
    call("srhip_reduce_conv_wgrad", _p(part), _p(cs), S, _p(dW), _p(db), Cout, Cin, _st())


class _ConvWgradItem(ctypes.Structure):   # srhip_conv_wgrad_item (include/srhip.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p)]


def conv3x3_wgrad_batched(items):
    """items: sequence of (dY NHWC [B,H,W,Cout], X NHWC [B,H,W,Cin], dW [Cout,Cin,3,3], db [Cout]) of ONE
    shape and dense NHWC layout -> all weight / bias gradients by one contraction + one reducer launch
    (bf16x3; up to 40 items per call)."""
    dY0, X0 = items[0][0], items[0][1]
    B, H, W, Cout = dY0.shape
    Cin = X0.shape[3]
    assert bx3_for(Cout, Cin), "batched conv weight gradients run on the bf16x3 kernels (>= 64 channels)"
    for j0 in range(0, len(items), 40):
        chunk = items[j0:j0 + 40]
        n = len(chunk)
        S = ctypes.c_int(0)
        per = ctypes.c_long(0)
        call("srhip_conv3x3_wgrad_batched_plan", n, B, H, W, Cout, Cin, ctypes.addressof(S), ctypes.addressof(per))
        part = SCRATCH.get("tnb_part", n * per.value, device=dY0.device)
        cs = SCRATCH.get("tnb_colsum", n * S.value * Cout, device=dY0.device)
        arr = (_ConvWgradItem * n)()
        for k, (dY, X, dW, db) in enumerate(chunk):
            _chk(dY, X, dW, db)
            assert dY.shape == dY0.shape and X.shape == X0.shape and dY.stride(2) == dY0.stride(2) \
                and X.stride(2) == X0.stride(2) and dY.is_contiguous() and X.is_contiguous()
            arr[k].dY, arr[k].X, arr[k].dW, arr[k].db = _p(dY), _p(X), _p(dW), _p(db)

        def run():
            call("srhip_conv3x3_wgrad_batched_bx3", ctypes.addressof(arr), n, dY0.stride(2), X0.stride(2), B, H, W,
                 Cout, Cin, _p(part), _p(cs), S.value, _st())
        if probe.on("conv_tn"):
            T = B * H * W
            with probe.timed(("conv_tn", f"batched x{n}", T, Cout, Cin), 18.0 * T * Cout * Cin * n,
                             4.0 * n * (T * Cin + T * Cout + 9 * Cin * Cout)):
                run()
        else:
            run()


# ------------------------------------------------------------------ weight prep
def fold_layernorm(W, b, gamma, beta, Wf, bf):
    _chk(W, b, gamma, beta, Wf, bf)
    call("srhip_fold_layernorm", _p(W), _p(b), _p(gamma), _p(beta), _p(Wf), _p(bf),
         W.shape[0], W.shape[1], _st())


def transpose(src, dst):
    _chk(src, dst)
    call("srhip_transpose", _p(src), _p(dst), src.shape[0], src.shape[1], _st())


def pack_conv_weight(w, wp=None, wpt=None):
    _chk(w, wp, wpt)
    call("srhip_pack_conv_weight", _p(w), _p(wp), _p(wpt), w.shape[0], w.shape[1], _st())


# ------------------------------------------------------------------ layernorm
def layernorm_fwd(x, stats=None, y=None, gamma=None, beta=None):
    _chk(x, stats, y, gamma, beta)
    C = x.shape[-1]
    M = x.numel() // C
    call("srhip_layernorm_fwd", _p(x), _p(stats), _p(y), _p(gamma), _p(beta), M, C, _st())


def layernorm_bwd(dy, x, stats, out, res=None, gamma=None, dgamma=None, dbeta=None):
    _chk(dy, x, stats, out, res, gamma, dgamma, dbeta)
    C = x.shape[-1]
    M = x.numel() // C
    ws = None
    if gamma is not None:
        ws = SCRATCH.get("ln_bwd_ws", lib.srhip_layernorm_bwd_ws(M, C), device=x.device)
    call("srhip_layernorm_bwd", _p(dy), _p(x), _p(stats), _p(res), _p(gamma), _p(out),
         _p(dgamma), _p(dbeta), _p(ws), M, C, _st())


# ------------------------------------------------------------------ attention
def bias_expand(table, biasT, biasN):
    _chk(table, biasT, biasN)
    call("srhip_bias_expand", _p(table), _p(biasT), _p(biasN), table.shape[1], _st())


def bias_grad(dbiasT, dtable):
    _chk(dbiasT, dtable)
    call("srhip_bias_grad", _p(dbiasT), _p(dtable), dtable.shape[1], _st())


def bias_grad_batched(dbiasT_all, first, dtables):
    """bias_grad for the consecutive attention blocks first .. first+len(dtables)-1 of one image buffer
    [nblocks, heads, 64, 64] in one launch (<= 8 blocks)."""
    _chk(dbiasT_all, *dtables)
    n = len(dtables)
    arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dtables])
    img = dbiasT_all[first]
    call("srhip_bias_grad_batched", _p(img), dbiasT_all.stride(0), ctypes.addressof(arr), n,
         dtables[0].shape[1], _st())


# the attention core on two fp16 planes / three products (wattn2.hip); SRHIP_WATTN_F16=0: the exact-f32 MFMA kernels
WATTN_F16 = _os.environ.get("SRHIP_WATTN_F16", "1") != "0"


def wattn_f16_ok(C, heads):
    return WATTN_F16 and use_bx3() and C % heads == 0 and C // heads in (10, 16, 30, 32) and C % 2 == 0


def bias_expand_f16(table, biasF, biasG=None):
    _chk(table, biasF, biasG)
    call("srhip_bias_expand_f16x2", _p(table), _p(biasF), _p(biasG), table.shape[1], _st())


def wattn_dbias_ws(B, H, W, heads):
    """floats of the partial bias-gradient tiles one fp16x2 attention backward leaves behind"""
    return lib.srhip_window_attention_bwd_f16x2_ws(B, H, W, heads)


def wattn_dbias_reduce_f16(parts, dbiasT_all, first, B, H, W, heads):
    """The bias-gradient images of the blocks first .. first + len(parts) - 1 of dbiasT_all [nblocks, heads, 64, 64]
    from the partial tiles parts[i] (rows of one [n, ws] buffer) their backward launches left, in ONE launch."""
    _chk(parts, dbiasT_all)
    assert parts.dim() == 2 and parts.stride(1) == 1 and dbiasT_all.shape[1] == heads
    call("srhip_window_attention_dbias_reduce_f16x2", _p(parts), parts.stride(0), parts.shape[0],
         _p(dbiasT_all[first]), dbiasT_all.stride(0), B, H, W, heads, _st())


def window_attention_bwd_f16(qkv, dout, dqkv, biasF, biasG, dbiasT, B, H, W, C, heads, shift, parts=None):
    """parts (a row of the buffer wattn_dbias_reduce_f16 takes): leave the partial bias-gradient tiles there instead
    of reducing them into dbiasT (which is then not written)."""
    _chk(qkv, dout, dqkv, biasF, biasG, dbiasT, parts)
    if parts is not None:
        ws, dbiasT = parts, None
    else:
        ws = SCRATCH.get("wattn2_ws", lib.srhip_window_attention_bwd_f16x2_ws(B, H, W, heads), device=qkv.device)
    args = (_p(qkv), _p(dout), _p(dqkv), _p(biasF), _p(biasG), _p(dbiasT), _p(ws), B, H, W, C, heads, shift, _st())
    if probe.on("wattn"):
        T = B * H * W
        with probe.timed(("wattn", "bwd", T, C), 10.0 * T * 64 * C, 4.0 * T * 7 * C):
            call("srhip_window_attention_bwd_f16x2", *args)
        return
    call("srhip_window_attention_bwd_f16x2", *args)


def window_attention_fwd_f16(qkv, out, biasF, B, H, W, C, heads, shift):
    _chk(qkv, out, biasF)
    args = (_p(qkv), _p(out), _p(biasF), B, H, W, C, heads, shift, _st())
    if probe.on("wattn"):
        T = B * H * W
        with probe.timed(("wattn", "fwd", T, C), 4.0 * T * 64 * C, 4.0 * T * 4 * C):
            call("srhip_window_attention_fwd_f16x2", *args)
        return
    call("srhip_window_attention_fwd_f16x2", *args)


def window_attention_fwd(qkv, out, biasT, B, H, W, C, heads, shift):
    _chk(qkv, out, biasT)
    if probe.on("wattn"):
        T = B * H * W
        with probe.timed(("wattn", "fwd", T, C), 4.0 * T * 64 * C, 4.0 * T * 4 * C):
            call("srhip_window_attention_fwd", _p(qkv), _p(out), _p(biasT), B, H, W, C, heads, shift, _st())
        return
    call("srhip_window_attention_fwd", _p(qkv), _p(out), _p(biasT), B, H, W, C, heads, shift, _st())


def window_attention_bwd(qkv, dout, dqkv, biasT, biasN, dbiasT, B, H, W, C, heads, shift):
    _chk(qkv, dout, dqkv, biasT, biasN, dbiasT)
    ws = SCRATCH.get("wattn_ws", lib.srhip_window_attention_bwd_ws(B, H, W, heads), device=qkv.device)
    args = (_p(qkv), _p(dout), _p(dqkv), _p(biasT), _p(biasN), _p(dbiasT), _p(ws), B, H, W, C, heads, shift, _st())
    if probe.on("wattn"):
        T = B * H * W
        with probe.timed(("wattn", "bwd", T, C), 10.0 * T * 64 * C, 4.0 * T * 7 * C):
            call("srhip_window_attention_bwd", *args)
        return
    call("srhip_window_attention_bwd", *args)


# ------------------------------------------------------------------ edge convs
def relu_mask(g, a):
    """g *= (a > 0) in place (backward of a ReLU whose output a was kept)."""
    _chk(g, a)
    call("srhip_relu_mask", _p(g), _p(a), g.numel(), _st())
    return g


def nearest_up2(lo, hi, adjoint=False):
    """NHWC nearest-neighbour x2: hi [B,2h,2w,C] = each lo pixel four times; adjoint: lo = sum of the four copies."""
    _chk(lo, hi)
    B, h, w, C = lo.shape
    assert hi.shape == (B, 2 * h, 2 * w, C) and lo.is_contiguous() and hi.is_contiguous()
    call("srhip_nearest_up2_nhwc", _p(lo), _p(hi), B, h, w, C, int(bool(adjoint)), _st())
    return lo if adjoint else hi


def leaky_relu_(x, alpha):
    """x = x > 0 ? x : alpha * x in place."""
    _chk(x)
    call("srhip_leaky_relu", _p(x), x.numel(), float(alpha), _st())
    return x


def leaky_relu_mask(g, a, alpha):
    """g *= (a > 0 ? 1 : alpha) in place (backward of a LeakyReLU whose output a was kept)."""
    _chk(g, a)
    call("srhip_leaky_relu_mask", _p(g), _p(a), g.numel(), float(alpha), _st())
    return g


def conv3x3_cin1_fwd(x, w, bias, Co, out=None, flip=False, relu=False):
    """x [B,H,W] -> NHWC [B,H,W,Co]; w torch layout [Co,1,3,3] (or [1,Co,3,3] with flip)."""
    flip = int(bool(flip)) | (2 if relu else 0)
    _chk(x, w, bias, out)
    B, H, W = x.shape
    if out is None:
        out = torch.empty(B, H, W, Co, device=x.device, dtype=torch.float32)
    call("srhip_conv3x3_cin1_fwd", _p(x), _p(w), _p(bias), _p(out), out.stride(2), B, H, W, Co,
         int(flip), _st())
    return out


def conv3x3_cin1_wgrad(x, dy, dw, db, flip=False):
    _chk(x, dy, dw, db)
    B, H, W = x.shape
    Co = dy.shape[3]
    ws = SCRATCH.get("cin1_ws", lib.srhip_conv3x3_cin1_wgrad_ws(Co), device=x.device)
    call("srhip_conv3x3_cin1_wgrad", _p(x), _p(dy), dy.stride(2), _p(dw), _p(db), _p(ws), B, H, W,
         Co, int(flip), _st())


def conv3x3_cout1_fwd(x, w, bias, out=None):
    """x NHWC [B,H,W,Ci] -> [B,H,W]; w torch layout [1,Ci,3,3]."""
    _chk(x, w, bias, out)
    B, H, W, Ci = x.shape
    if out is None:
        out = torch.empty(B, H, W, device=x.device, dtype=torch.float32)
    call("srhip_conv3x3_cout1_fwd", _p(x), x.stride(2), _p(w), _p(bias), _p(out), B, H, W, Ci, _st())
    return out


# ------------------------------------------------------------------ pixel shuffle
def pixel_shuffle_add_ok(Co, r):
    """shapes srhip_pixel_shuffle_add takes"""
    return r > 1 and Co * (r * r + 1) * 4 <= 48 * 1024 and Co * r * r >= 256


def pixel_shuffle(x, r, nhwc_out=False, inverse=False, out=None, add=None, fac=1.0):
    """forward: x NHWC [B,h,w,Co*r*r] -> NCHW [B,Co,h*r,w*r] (or NHWC).  inverse:
    x is the high-res side and the NHWC low-res tensor is returned.  add (NHWC forward only): out = shuffled + fac * add."""
    _chk(x, out, add)
    if add is not None:
        B, h, w, C = x.shape
        Co = C // (r * r)
        assert nhwc_out and not inverse and add.is_contiguous() and tuple(add.shape) == (B, h * r, w * r, Co)
        if out is None:
            out = torch.empty(B, h * r, w * r, Co, device=x.device, dtype=torch.float32)
        call("srhip_pixel_shuffle_add", _p(x), _p(out), B, h, w, Co, r, _p(add), float(fac), _st())
        return out
    if not inverse:
        B, h, w, C = x.shape
        Co = C // (r * r)
        if out is None:
            shape = (B, h * r, w * r, Co) if nhwc_out else (B, Co, h * r, w * r)
            out = torch.empty(shape, device=x.device, dtype=torch.float32)
    else:
        if nhwc_out:
            B, Hh, Ww, Co = x.shape
        else:
            B, Co, Hh, Ww = x.shape
        h, w = Hh // r, Ww // r
        if out is None:
            out = torch.empty(B, h, w, Co * r * r, device=x.device, dtype=torch.float32)
    call("srhip_pixel_shuffle", _p(x), _p(out), B, h, w, Co, r, int(nhwc_out), int(inverse), _st())
    return out


# ------------------------------------------------------------------ losses
def loss_l1l2(pred, target, mode, lam=1.0, weight=None, grad=None, loss_out=None,
              grad_accum=False, loss_accum=False):
    _chk(pred, target, weight, grad, loss_out)
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("loss_ws", 2048, torch.float64, pred.device)
    call("srhip_loss_l1l2", _p(pred), _p(target), _p(weight), _p(grad), _p(loss_out), _p(ws),
         pred.numel(), mode, float(lam), int(grad_accum), int(loss_accum), _st())
    return loss_out


def ssim_loss(pred, target, ws, lam=1.0, grad=None, loss_out=None, grad_accum=False,
              loss_accum=False):
    """pred/target [B,1,H,W] or [B,H,W]."""
    _chk(pred, target, grad, loss_out)
    B = pred.shape[0]
    H, W = pred.shape[-2:]
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    wsb = SCRATCH.get("ssim_ws", lib.srhip_ssim_loss_ws(B, H, W), device=pred.device)
    call("srhip_ssim_loss", _p(pred), _p(target), _p(grad), _p(loss_out), _p(wsb), B, H, W, ws,
         float(lam), int(grad_accum), int(loss_accum), _st())
    return loss_out


def loss_pointwise(pred, target, mode, lam=1.0, eps=1e-9, weight=None, grad=None, loss_out=None,
                   grad_accum=False, loss_accum=False):
    """mode 0 L1 | 1 L2 | 2 Charbonnier(eps) | 3 L2Sum (dlib/loss/main.py:45-151)."""
    _chk(pred, target, weight, grad, loss_out)
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("loss_ws", 2048, torch.float64, pred.device)
    call("srhip_loss_pointwise", _p(pred), _p(target), _p(weight), _p(grad), _p(loss_out), _p(ws),
         pred.numel(), mode, float(lam), float(eps), int(grad_accum), int(loss_accum), _st())
    return loss_out


def loss_bounded(pred, target, lam=1.0, eps=0.0, t=1.0, scale=1.0, grad=None, loss_out=None,
                 grad_accum=False, loss_accum=False):
    """BoundedPrediction through the extended log barrier at parameter t (dlib/loss/main.py:189-237,
    dlib/losses/elb.py:92-122); scale = color_max with restore_range."""
    _chk(pred, target, grad, loss_out)
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("loss_ws", 2048, torch.float64, pred.device)
    call("srhip_loss_bounded", _p(pred), _p(target), _p(grad), _p(loss_out), _p(ws), pred.numel(),
         float(lam), float(eps), float(t), float(scale), int(grad_accum), int(loss_accum), _st())
    return loss_out


def l1_sparsity(w, lam=1.0, grad=None, loss_out=None, loss_accum=False):
    """WeightsSparsityLoss on a flat parameter range (dlib/loss/main.py:938-959): loss_out (+)= lam*sum|w|,
    grad += lam*sign(w)."""
    _chk(w, grad, loss_out)
    if loss_out is None:
        loss_out = torch.empty(1, device=w.device, dtype=torch.float32)
    ws = SCRATCH.get("loss_ws", 2048, torch.float64, w.device)
    call("srhip_l1_sparsity", _p(w), _p(grad), _p(loss_out), _p(ws), w.numel(), float(lam), int(loss_accum), _st())
    return loss_out


def loss_local_moments(pred, target, lam=1.0, grad=None, loss_out=None, grad_accum=False, loss_accum=False):
    """LocalMoments on 1-channel images (dlib/loss/main.py:240-325; fixed 3x3 window, reflect padding)."""
    _chk(pred, target, grad, loss_out)
    assert pred.ndim == 3 or pred.shape[1] == 1, "LocalMoments: 1-channel images (local_terms.py:36)"
    B = pred.shape[0]
    H, W = pred.shape[-2:]
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("stencil_ws", lib.srhip_loss_stencil_ws(B, H, W), torch.float64, pred.device)
    call("srhip_loss_local_moments", _p(pred), _p(target), _p(grad), _p(loss_out), _p(ws), B, H, W, float(lam),
         int(grad_accum), int(loss_accum), _st())
    return loss_out


def loss_hist(pred, target, lam=1.0, norm=2, sigma=1e5, bins=256, grad=None, loss_out=None, grad_accum=False,
              loss_accum=False, elb_t=1.0):
    """HistogramMatch over soft histograms of each image of the batch (dlib/loss/main.py:690-782,
    dlib/loss/global_terms.py:17-72); norm 1 | 2 | 3 (KL) | 4 (Bhattacharyya through the log barrier at elb_t)."""
    _chk(pred, target, grad, loss_out)
    B = pred.shape[0]
    n = pred.numel() // B
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("hist_ws", lib.srhip_loss_hist_ws(B, bins), device=pred.device)
    call("srhip_loss_hist", _p(pred), _p(target), _p(grad), _p(loss_out), _p(ws), B, n, int(bins), float(sigma),
         int(norm), float(lam), int(grad_accum), int(loss_accum), float(elb_t), _st())
    return loss_out


def loss_kde(pred, target, lam=1.0, norm=2, kde_bw=1. / 255. ** 2, bins=256, grad=None, loss_out=None,
             grad_accum=False, loss_accum=False, elb_t=1.0):
    """KDEMatch over a Gaussian KDE of each 1-channel image in [0, 1] (dlib/loss/main.py:785-898,
    dlib/loss/global_terms.py:75-152); norm 1 | 2 | 4 (Bhattacharyya through the log barrier at elb_t)."""
    _chk(pred, target, grad, loss_out)
    assert pred.ndim == 3 or pred.shape[1] == 1, "KDEMatch: ndim == 1 (main.py:815)"
    B = pred.shape[0]
    n = pred.numel() // B
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("hist_ws", lib.srhip_loss_hist_ws(B, bins), device=pred.device)
    call("srhip_loss_kde", _p(pred), _p(target), _p(grad), _p(loss_out), _p(ws), B, n, int(bins), float(kde_bw),
         int(norm), float(lam), int(grad_accum), int(loss_accum), float(elb_t), _st())
    return loss_out


STENCIL_OPS = {"grad": 0, "laplace": 1, "lv": 2}


def loss_stencil(pred, target, kind, lam=1.0, norm=2, ksz=3, channel_norm=False, grad=None,
                 loss_out=None, grad_accum=False, loss_accum=False):
    """Local-variation loss family on 1-channel images [B,1,H,W] / [B,H,W]
    (dlib/loss/main.py:328-674): kind 'grad' | 'laplace' | 'lv'."""
    _chk(pred, target, grad, loss_out)
    assert pred.ndim == 3 or pred.shape[1] == 1, "local-variation losses: 1-channel images (local_variations.py:45)"
    B = pred.shape[0]
    H, W = pred.shape[-2:]
    if loss_out is None:
        loss_out = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = SCRATCH.get("stencil_ws", lib.srhip_loss_stencil_ws(B, H, W), torch.float64, pred.device)
    call("srhip_loss_stencil", _p(pred), _p(target), _p(grad), _p(loss_out), _p(ws), B, H, W,
         STENCIL_OPS[kind], int(ksz), int(norm), int(channel_norm), float(lam), int(grad_accum),
         int(loss_accum), _st())
    return loss_out


# ------------------------------------------------------------------ input pipeline (f2)
class _PatchJob(ctypes.Structure):   # srhip_patch_job (include/srhip.h)
    _fields_ = [("img", ctypes.c_void_p), ("H", ctypes.c_int), ("W", ctypes.c_int), ("y0", ctypes.c_int),
                ("x0", ctypes.c_int), ("mode", ctypes.c_int)]


def patch_gather(tiles, ids, y0, x0, modes, P, out=None):
    """[B,1,P,P] float32 batch out of device-resident uint8 tiles: crop at (y0, x0), augment_img mode 0..7,
    /255 -- the per-sample tail of DatasetDPSR.__getitem__ (dataset_dpsr.py:866-894,914-915).  ``tiles``:
    sequence of contiguous uint8 CUDA tensors [H, W]; ids / y0 / x0 / modes: per-sample Python ints."""
    B = len(ids)
    dev = tiles[ids[0]].device
    if out is None:
        out = torch.empty(B, 1, P, P, device=dev, dtype=torch.float32)
    _chk(out)
    jobs = (_PatchJob * B)()
    for b in range(B):
        t = tiles[ids[b]]
        if not (t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous()):
            raise ValueError("patch_gather: tiles must be contiguous uint8 CUDA tensors [H, W]")
        jobs[b].img, jobs[b].H, jobs[b].W = t.data_ptr(), t.shape[0], t.shape[1]
        jobs[b].y0, jobs[b].x0, jobs[b].mode = int(y0[b]), int(x0[b]), int(modes[b])
    call("srhip_patch_gather", ctypes.addressof(jobs), B, P, _p(out), _st())
    return out


def resize_cubic(src, size_hw, out=None):
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_CUBIC) on a batch of 1-channel images [B, H, W] (uint8 or float32)
    -> [B, h, w] of the same dtype (resize.hip; dataset_dpsr.py:659-683)."""
    if not src.is_cuda:
        raise SrhipError("srhip ops need CUDA/HIP tensors (no CPU fallback exists)")
    assert src.dim() == 3 and src.is_contiguous() and src.dtype in (torch.uint8, torch.float32), (src.shape, src.dtype)
    B, H, W = src.shape
    Ho, Wo = int(size_hw[0]), int(size_hw[1])
    if out is None:
        out = torch.empty(B, Ho, Wo, device=src.device, dtype=src.dtype)
    call("srhip_resize_cubic", _p(src), _p(out), int(src.dtype == torch.uint8), B, H, W, Ho, Wo, _st())
    return out


def u8_to_unit(src, out=None):
    """uint8 -> float32 / 255 (util.uint2single)."""
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    call("srhip_u8_to_unit", _p(src), _p(out), src.numel(), _st())
    return out


def clip01_(x):
    call("srhip_clip01", _p(x), x.numel(), _st())
    return x


def im2col_c1(x, ksize, ldo, out=None):
    """x [B,H,W] -> patch matrix [B*H*W, ldo] of a ksize x ksize / pad ksize//2 convolution (zero padded)."""
    _chk(x, out)
    B, H, W = x.shape
    if out is None:
        out = torch.empty(B * H * W, ldo, device=x.device, dtype=torch.float32)
    call("srhip_im2col_c1", _p(x), _p(out), ldo, B, H, W, int(ksize), _st())
    return out


def roi_sample(tiles, ids, P, threshold, uniforms):
    """Patch origins [B,2] (row, col; int32, on the device) drawn with the reference's ROI weighting
    (PatchSampler._roi, dataset_dpsr.py:330-369) from ``uniforms`` (float64 [B] in [0,1), e.g. torch.rand on the
    device): the sampling step of the training crops without a host round trip."""
    B = len(ids)
    dev = tiles[ids[0]].device
    if not (uniforms.is_cuda and uniforms.dtype == torch.float64 and uniforms.numel() == B):
        raise ValueError("roi_sample: uniforms must be a float64 CUDA tensor with one entry per patch")
    jobs = (_PatchJob * B)()
    rows = 1
    for b in range(B):
        t = tiles[ids[b]]
        if not (t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous()):
            raise ValueError("roi_sample: tiles must be contiguous uint8 CUDA tensors [H, W]")
        jobs[b].img, jobs[b].H, jobs[b].W = t.data_ptr(), t.shape[0], t.shape[1]
        rows = max(rows, t.shape[0] - P)
    ws = SCRATCH.get("roi_ws", lib.srhip_roi_sample_ws(B, rows), torch.int32, dev)
    out = torch.empty(B, 2, dtype=torch.int32, device=dev)
    call("srhip_roi_sample", ctypes.addressof(jobs), B, P, int(threshold), _p(uniforms.contiguous()), _p(ws), rows,
         _p(out), _st())
    return out


def train_batch(hr_tiles, lr_tiles, ids, y0, x0, modes, patch_size, sf):
    """The batch dict the reference's trainer feeds ModelPlain (keys l_im, h_im; dataset_dpsr.py:981-1005)
    from resident tiles: HR crop at (y0, x0), LR crop at (y0 // sf, x0 // sf) of size patch_size // sf,
    the same augmentation mode on both (dataset_dpsr.py:866-894)."""
    return {"h_im": patch_gather(hr_tiles, ids, y0, x0, modes, patch_size),
            "l_im": patch_gather(lr_tiles, ids, [v // sf for v in y0], [v // sf for v in x0], modes,
                                 patch_size // sf)}


# ------------------------------------------------------------------ metrics
def tensor2uint82float(x):
    _chk(x)
    x = x.float().contiguous()
    out = torch.empty_like(x)
    call("srhip_tensor2uint82float", _p(x), _p(out), x.numel(), _st())
    return out


def metrics_psnr_family(E, Hh, border, thresholds=(), inputs_are_u8=False):
    """Returns fp64 [B, 1+len(thresholds), 4] = PSNR, PSNR_Y, MSE, NRMSE."""
    _chk(E, Hh)
    B = E.shape[0]
    H, W = E.shape[-2:]
    nth = len(thresholds)
    th = torch.tensor(list(thresholds) or [0], dtype=torch.int32, device=E.device)
    ws = SCRATCH.get("met_ws", lib.srhip_metrics_ws(B, nth), torch.float64, E.device)
    out = torch.empty(B, nth + 1, 4, dtype=torch.float64, device=E.device)
    call("srhip_metrics_psnr_family", _p(E), _p(Hh), B, H, W, border, _p(th), nth,
         int(inputs_are_u8), _p(ws), _p(out), _st())
    return out


def metrics_ssim(E, Hh, border, thresholds=(), inputs_are_u8=False):
    _chk(E, Hh)
    B = E.shape[0]
    H, W = E.shape[-2:]
    nth = len(thresholds)
    th = torch.tensor(list(thresholds) or [0], dtype=torch.int32, device=E.device)
    ws = SCRATCH.get("ssim_met_ws", 2 * B * (nth + 1), torch.float64, E.device)
    out = torch.empty(B, nth + 1, dtype=torch.float32, device=E.device)
    call("srhip_metrics_ssim", _p(E), _p(Hh), B, H, W, border, _p(th), nth, int(inputs_are_u8),
         _p(ws), _p(out), _st())
    return out


# ------------------------------------------------------------------ optimizers
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, gscale=1.0, skip_flag=None):
    _chk(p, g, m, v, skip_flag)
    call("srhip_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), step, float(lr), float(b1),
         float(b2), float(eps), float(wd), float(gscale), _p(skip_flag), _st())


def sgd_step(p, g, buf, lr, momentum=0.9, wd=0.0, nesterov=True, first=False, gscale=1.0,
             skip_flag=None):
    _chk(p, g, buf, skip_flag)
    call("srhip_sgd_step", _p(p), _p(g), _p(buf), p.numel(), float(lr), float(momentum),
         float(wd), int(nesterov), int(first), float(gscale), _p(skip_flag), _st())


def optim_tick(skip_flag, counter):
    """counter += (skip_flag == 0) on the device (steps actually applied)."""
    _chk(skip_flag, counter)
    call("srhip_optim_tick", _p(skip_flag), _p(counter), _st())


def adam_step_dc(p, g, m, v, counter, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, gscale=1.0, skip_flag=None,
                 lr_dev=None):
    _chk(p, g, m, v, counter, skip_flag, lr_dev)
    call("srhip_adam_step_dc", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(counter), float(lr), float(b1),
         float(b2), float(eps), float(wd), float(gscale), _p(skip_flag), _p(lr_dev), _st())


def sgd_step_dc(p, g, buf, counter, lr, momentum=0.9, wd=0.0, nesterov=True, gscale=1.0, skip_flag=None,
                lr_dev=None):
    _chk(p, g, buf, counter, skip_flag, lr_dev)
    call("srhip_sgd_step_dc", _p(p), _p(g), _p(buf), p.numel(), _p(counter), float(lr), float(momentum),
         float(wd), int(nesterov), float(gscale), _p(skip_flag), _p(lr_dev), _st())


def grad_norm_clip(g, gscale, max_norm, norm_coef):
    """clip_grad_norm_(max_norm, 2) over the flat gradient (model_plain.py:350-361): norm_coef[0] = ||g * gscale||_2,
    norm_coef[1] = min(1, max_norm / (norm + 1e-6)), g *= norm_coef[1].  Device-resident, deterministic, capturable."""
    _chk(g, norm_coef)
    assert norm_coef.numel() >= 2 and norm_coef.dtype == torch.float32
    wsb = lib.srhip_grad_norm_clip_ws()
    ws = SCRATCH.get("clip_ws", wsb // 8, torch.float64, g.device)
    call("srhip_grad_norm_clip", _p(g), g.numel(), float(gscale), float(max_norm), _p(norm_coef), _p(ws), wsb, _st())


def ema_update(e, p, decay, skip_flag=None):
    """e = e * decay + p * (1 - decay) (ModelBase.update_E, model_base.py:213-219); skipped on the device if *skip_flag."""
    _chk(e, p, skip_flag)
    assert e.numel() == p.numel()
    call("srhip_ema_update", _p(e), _p(p), e.numel(), float(decay), _p(skip_flag), _st())


def nonfinite_flag(x, flag):
    _chk(x, flag)
    call("srhip_nonfinite_flag", _p(x), x.numel(), _p(flag), _st())


def axpby(y, x, a, b):
    _chk(x, y)
    call("srhip_axpby", _p(y), _p(x), y.numel(), float(a), float(b), _st())


def sum_into(x, out):
    """out[0] = sum(x)."""
    _chk(x, out)
    ws = SCRATCH.get("loss_ws", 2048, torch.float64, x.device)
    call("srhip_sum", _p(x), x.numel(), _p(out), _p(ws), _st())


# ------------------------------------------------------------------ BatchNorm2d (channels-last)
def _bn_ws(T, C, device):
    nb = ctypes.c_long(0)
    call("srhip_bn_workspace_bytes", T, C, ctypes.addressof(nb))
    return SCRATCH.get("bn_ws", (nb.value + 7) // 8, torch.float64, device), nb.value


def bn_stats(x, gamma, beta, coef, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """Training statistics of x [..., C] (dense, channels last) -> coef [4, C] = mean, rstd, gamma * rstd, beta;
    running statistics updated in place when given (nn.BatchNorm2d semantics)."""
    _chk(x, gamma, beta, coef, running_mean, running_var)
    C = gamma.numel()
    T = x.numel() // C
    assert x.is_contiguous() and coef.shape == (4, C) and coef.is_contiguous()
    ws, nb = _bn_ws(T, C, x.device)
    call("srhip_bn_stats", _p(x), T, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(momentum),
         float(eps), _p(coef), _p(ws), nb, _st())
    return coef


def bn_apply(x, coef, out=None, relu=True):
    """out = relu?((x - mean) * k + beta) with coef [4, C] (bn_stats, or the running statistics in eval mode)."""
    _chk(x, coef, out)
    C = coef.shape[1]
    if out is None:
        out = torch.empty_like(x)
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
    call("srhip_bn_apply", _p(x), _p(coef), _p(out), x.numel() // C, C, int(bool(relu)), _st())
    return out


def bn_bwd(dy, x, coef, a=None, dx=None, res=None, dgamma=None, dbeta=None, accumulate=False):
    """Backward of relu?(BN(x)) in training mode: dz = dy * (a > 0) if the ReLU output ``a`` is given; dgamma / dbeta are
    written, or added to with ``accumulate``; dx (if given) = the input gradient (+ res)."""
    _chk(dy, x, coef, a, dx, res, dgamma, dbeta)
    C = coef.shape[1]
    T = x.numel() // C
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape
    for t in (a, dx, res):
        assert t is None or (t.is_contiguous() and t.shape == x.shape)
    ws, nb = _bn_ws(T, C, x.device)
    call("srhip_bn_bwd", _p(dy), _p(a), _p(x), _p(coef), T, C, _p(dx), _p(res), _p(dgamma), _p(dbeta), int(bool(accumulate)),
         _p(ws), nb, _st())
    return dx
