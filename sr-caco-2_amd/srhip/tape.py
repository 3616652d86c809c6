"""A small tape engine for the conv-family networks of the reference's 16-method sweep (DBPN, SRFBN,
ProSR -- dlib/models/network_dbpn.py, network_srfbn.py, network_prosr.py).

These nets are dense graphs (back-projection stages, feedback loops, dense blocks: concatenations, weight sharing
across iterations) of a handful of operations that libsrhip already has kernels for.  Instead of a hand-sequenced
engine per net (swinir_engine.py, edsr_engine.py ...), the network's forward is written ONCE as calls on a Tape; every
call launches its libsrhip kernels and records a backward closure; Tape.backward replays the closures in reverse.
No aten arithmetic: PyTorch supplies device memory, views and the index tensors that re-lay weights.

How the reference's layers map onto the kernels:
  * nn.Conv2d(k=3, s=1, p=1)                    -> the implicit-GEMM 3x3 conv (bf16x3 / fp16x2 planes from 64 channels on,
                                                   exact-f32 MFMA below; 1-channel ends: small.hip)
  * nn.Conv2d(k=1)                              -> the NT GEMM on the token matrix [B*H*W][C]
  * nn.ConvTranspose2d(k=s+4, stride s, p=2)    -> conv3x3 C -> C*s*s with zero-padded sub-kernels + PixelShuffle(s):
      out[y s + i][x s + j] = sum_{dy,dx} in[y + dy][x + dx] . Wt[:, :, i + 2 - dy s, j + 2 - dx s]
  * nn.Conv2d(k=s+4, stride s, p=2)             -> PixelUnshuffle(s) + conv3x3 C*s*s -> C with zero-padded sub-kernels:
      out[y][x] = sum_{dy,dx} unshuffled[y + dy][x + dx][(ci, i, j)] . W[:, ci, dy s + i + 2, dx s + j + 2]
  * nn.PReLU() / ReLU / LeakyReLU, torch.cat, add / sub, nn.PixelShuffle, nn.ReflectionPad2d(1)
Every tensor is NHWC.  Training buffers come from a pool keyed by shape with liveness (_Pool): a value's activation and
gradient return to it when the op that produced the value has run its backward; the request sequence repeats every step, so a
training step allocates nothing after the first one.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import call
from .swinir_engine import _Bufs


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


LIMIT = (1 << 29) - 1      # the kernels' 32-bit staging offsets: a tensor operand stays below 2^29 floats per call


def _chunks(B, per_image):
    """batch slices [b0, b1) whose largest operand (per_image floats per image) stays under LIMIT"""
    nb = max(1, min(B, LIMIT // max(1, per_image)))
    return [(b0, min(B, b0 + nb)) for b0 in range(0, B, nb)]


class Var:
    """A value on the tape: NHWC tensor `t` (1-channel images: [B, H, W]) and its gradient `g` (contiguous, same shape).
    `gpool`: g was taken from the tape's buffer pool (and goes back there once the op that produced the value has run its
    backward)."""
    __slots__ = ("t", "g", "need", "idx", "gpool")

    def __init__(self, t, need=True, idx=-1):
        self.t, self.g, self.need, self.idx, self.gpool = t, None, need, idx, False


class _Pool:
    """Training buffers of a tape net, by shape, with liveness (round 5).  Round 4 kept one persistent
    buffer per tape position and one gradient buffer per value for the whole step: every activation AND every gradient of
    the graph at once (SRFBN 234 GiB at the README batch, DBPN beyond 288).  A value's activation and gradient are dead as
    soon as the op that PRODUCED it has run its backward closure -- its consumers come later on the tape, so their
    closures have already run -- and go back to the pool there; gradient buffers of the ops in front are then served from
    the activations just released.  The peak is the forward's activations plus a handful of gradients.  The sequence of
    requests is the same in every step, so every request gets the same tensor again: nothing is allocated after the first
    step and a captured step (TrainStep.step_graph) stays valid."""

    def __init__(self):
        self.all, self.free = {}, {}
        self.used = set()

    def reset(self):
        """every buffer free again (a new forward; whatever a forward without a backward left marked as taken included).
        Shapes the LAST step never asked for (a ragged last batch, another patch size) are released here instead of staying
        resident beside the current set -- SRFBN / DBPN sit near the HBM capacity at the README batch; a captured
        step that held them is re-made (ops.note_realloc)."""
        if self.used:
            stale = [k for k in self.all if k not in self.used]
            if stale:
                for k in stale:
                    del self.all[k]
                ops.note_realloc()
        self.used = set()
        self.free = {k: list(v) for k, v in self.all.items()}

    def take(self, shape, device):
        key = tuple(shape)
        self.used.add(key)
        fl = self.free.get(key)
        if fl:
            return fl.pop()
        t = torch.empty(key, device=device)
        self.all.setdefault(key, []).append(t)
        self.free.setdefault(key, [])
        return t

    def give(self, t):
        self.free[tuple(t.shape)].append(t)

    def nbytes(self):
        return sum(4 * t.numel() for v in self.all.values() for t in v)


class _BackList(list):
    """The tape's backward closures.  append() also records which values and buffers were created since the previous
    append: they belong to the op whose closure this is, and are released when it has run."""

    def __init__(self, tape):
        super().__init__()
        self.tape = tape

    def append(self, fn):
        t = self.tape
        super().append((fn, t._recent_vars, t._recent_bufs))
        t._recent_vars, t._recent_bufs = [], []


# --------------------------------------------------------------------------- weight re-layout maps (host, cached)
_MAPS = {}


def _deconv_map(s, k, p, device):
    """ConvTranspose2d(k, stride s, padding p) as conv3x3 + PixelShuffle(s): for sub-pixel (i, j) and tap (dy, dx) the
    source kernel element (ky, kx) = (i + p - dy s, j + p - dx s), or none.  Returns (index [s, s, 3, 3] into k*k,
    valid [s, s, 3, 3])."""
    key = ("deconv", s, k, p, str(device))
    if key not in _MAPS:
        idx = np.zeros((s, s, 3, 3), dtype=np.int64)
        val = np.zeros((s, s, 3, 3), dtype=np.float32)
        for i in range(s):
            for j in range(s):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        ky, kx = i + p - dy * s, j + p - dx * s
                        if 0 <= ky < k and 0 <= kx < k:
                            idx[i, j, dy + 1, dx + 1] = ky * k + kx
                            val[i, j, dy + 1, dx + 1] = 1.0
        assert val.sum() == k * k, "every kernel element must appear exactly once (k = s + 2 p)"
        _MAPS[key] = (torch.from_numpy(idx).to(device), torch.from_numpy(val).to(device))
    return _MAPS[key]


def _down_map(s, k, p, device):
    """Conv2d(k, stride s, padding p) as PixelUnshuffle(s) + conv3x3: (ky, kx) = (dy s + i + p, dx s + j + p)."""
    key = ("down", s, k, p, str(device))
    if key not in _MAPS:
        idx = np.zeros((s, s, 3, 3), dtype=np.int64)
        val = np.zeros((s, s, 3, 3), dtype=np.float32)
        for i in range(s):
            for j in range(s):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        ky, kx = dy * s + i + p, dx * s + j + p
                        if 0 <= ky < k and 0 <= kx < k:
                            idx[i, j, dy + 1, dx + 1] = ky * k + kx
                            val[i, j, dy + 1, dx + 1] = 1.0
        assert val.sum() == k * k
        _MAPS[key] = (torch.from_numpy(idx).to(device), torch.from_numpy(val).to(device))
    return _MAPS[key]


def expand_deconv(wt, s, p):
    """ConvTranspose2d weight [Ci, Co, k, k] -> conv3x3 weight [Co*s*s, Ci, 3, 3] (PixelShuffle channel order)."""
    Ci, Co, k, _ = wt.shape
    idx, val = _deconv_map(s, k, p, wt.device)
    g = wt.reshape(Ci, Co, k * k)[:, :, idx.reshape(-1)].reshape(Ci, Co, s, s, 3, 3) * val
    return g.permute(1, 2, 3, 0, 4, 5).reshape(Co * s * s, Ci, 3, 3).contiguous()


def collapse_deconv(dwe, Ci, Co, s, k, p):
    """Gradient of expand_deconv: [Co*s*s, Ci, 3, 3] -> [Ci, Co, k, k] (every kernel element has exactly one image)."""
    idx, val = _deconv_map(s, k, p, dwe.device)
    g = dwe.reshape(Co, s, s, Ci, 3, 3).permute(3, 0, 1, 2, 4, 5).reshape(Ci, Co, -1)
    out = torch.zeros(Ci, Co, k * k, device=dwe.device)
    sel = val.reshape(-1) > 0
    out[:, :, idx.reshape(-1)[sel]] = g[:, :, sel]
    return out.reshape(Ci, Co, k, k)


def expand_down(w, s, p):
    """Conv2d(k, stride s) weight [Co, Ci, k, k] -> conv3x3 weight [Co, Ci*s*s, 3, 3] on the unshuffled image."""
    Co, Ci, k, _ = w.shape
    idx, val = _down_map(s, k, p, w.device)
    g = w.reshape(Co, Ci, k * k)[:, :, idx.reshape(-1)].reshape(Co, Ci, s, s, 3, 3) * val
    return g.reshape(Co, Ci * s * s, 3, 3).contiguous()


def collapse_down(dwe, Co, Ci, s, k, p):
    idx, val = _down_map(s, k, p, dwe.device)
    g = dwe.reshape(Co, Ci, -1)
    out = torch.zeros(Co, Ci, k * k, device=dwe.device)
    sel = val.reshape(-1) > 0
    out[:, :, idx.reshape(-1)[sel]] = g[:, :, sel]
    return out.reshape(Co, Ci, k, k)


# --------------------------------------------------------------------------- prepared weights
class ConvW:
    """Kernel-ready forms of one conv layer: `w3` the (possibly expanded) [Co, Ci, 3, 3] weight, its forward / data-
    gradient packs (f32 [9, Co, Ci] / [9, Ci, Co], or Bx3 planes), or `w1` [Co, Ci] (+ transpose) of a 1x1 conv."""
    __slots__ = ("kind", "s", "k", "p", "Co", "Ci", "w3", "wp", "wpt", "w1", "w1T", "bias", "use_planes")


class WeightBank:
    """Derived weights of every conv of a net, rebuilt after the parameters changed (TapeEngine.prepare).  From 64
    channels on both sides the forward / data-gradient operands are split-MFMA planes (two fp16 planes + block exponents
    where the kernels take them, three bf16 planes otherwise: ops.PrepTable, ONE launch for the whole net); below that,
    f32 packs for the exact-f32 MFMA kernels."""

    def __init__(self):
        self.d = {}
        self.bufs = _Bufs()
        self.ws = ops.WeightSet()
        self._jobs, self._table, self._sig = [], None, None

    def begin(self):
        self._jobs = []

    def conv(self, key, mod_weight, bias, kind, s=1, k=3, p=1):
        """kind: 'c3' | 'c1' | 'deconv' | 'down'."""
        e = self.d.get(key)
        if e is None:
            e = self.d[key] = ConvW()
            e.kind, e.s, e.k, e.p = kind, s, k, p
        dev = mod_weight.device
        w = mod_weight.data
        if kind == "c1":
            e.Co, e.Ci = w.shape[0], w.shape[1]
            e.use_planes = ops.bx3_nt_for(e.Co, e.Ci) and e.Ci % 4 == 0 and e.Co % 4 == 0
            e.bias = None if bias is None else bias.data
            w2 = w.reshape(e.Co, e.Ci)
            if e.use_planes:
                e.w1 = self.ws.planes(key + ".w", e.Co, e.Ci, dev)
                e.w1T = self.ws.planes(key + ".wT", e.Ci, e.Co, dev)
                self._jobs.append(("lin", w2, e.w1, e.w1T))
            else:
                e.w1 = w2
                e.w1T = self.bufs.get(key + ".w1T", e.Ci, e.Co, device=dev)
                ops.transpose(w2.contiguous(), e.w1T)
            return e
        if kind == "c3":
            w3, b3 = w, (None if bias is None else bias.data)
        else:
            ex = expand_deconv(w, s, p) if kind == "deconv" else expand_down(w, s, p)
            w3 = self.bufs.get(key + ".w3", *ex.shape, device=dev)      # persistent: the preparation table holds its address
            w3.copy_(ex)
            b3 = None if bias is None else (bias.data.repeat_interleave(s * s).contiguous() if kind == "deconv" else bias.data)
        e.w3, e.bias = w3, b3
        e.Co, e.Ci = w3.shape[0], w3.shape[1]
        e.use_planes = ops.bx3_nt_for(e.Co, e.Ci) and e.Ci % 4 == 0 and e.Co % 4 == 0
        if e.use_planes:
            e.wp = self.ws.planes(key + ".wp", 9 * e.Co, e.Ci, dev)
            e.wpt = self.ws.planes(key + ".wpt", 9 * e.Ci, e.Co, dev)
            self._jobs.append(("conv", w3, e.wp, e.wpt))
        else:
            e.wp = self.bufs.get(key + ".wp", 9, e.Co, e.Ci, device=dev)
            e.wpt = self.bufs.get(key + ".wpt", 9, e.Ci, e.Co, device=dev)
            ops.pack_conv_weight(w3, e.wp, e.wpt)
        return e

    def finish(self, device):
        """one preparation launch for every split operand of the net"""
        if not self._jobs:
            return
        sig = tuple(j[1].data_ptr() for j in self._jobs)
        if self._table is None or sig != self._sig:
            tb = ops.PrepTable()
            for kind, w, fwd, bwd in self._jobs:
                if kind == "lin":
                    tb.linear(w, fwd)
                    tb.linear(w, bwd, transpose=True)
                else:
                    tb.conv(w, fwd)
                    tb.conv(w, bwd, data_grad=True)
            self._table, self._sig = tb.build(device), sig
        self._table.run()


# --------------------------------------------------------------------------- the tape
class Tape:
    def __init__(self, bufs, bank, save, device):
        self.bufs, self.bank, self.save, self.dev = bufs, bank, save, device
        self.n = 0
        self._recent_vars, self._recent_bufs = [], []
        self.back = _BackList(self)          # (backward closure, values, buffers of its op), forward order
        self.tag = "t" if save else "e"
        self._gtmp = 0
        self.pool = None
        if save:
            if getattr(bufs, "pool", None) is None:
                bufs.pool = _Pool()
            self.pool = bufs.pool
            self.pool.reset()

    # ---- buffers
    def new(self, *shape):
        """Output buffer of the op at this tape position.  Training keeps every one (persistent, keyed by the position: a
        step allocates nothing after the first one); inference takes a fresh tensor from torch's caching allocator, which
        hands the memory back as soon as the value has no consumer left -- a dense 70-layer graph at 512 x 512 would
        otherwise hold every activation it ever made."""
        self.n += 1
        if not self.save:
            return torch.empty(shape, device=self.dev)
        t = self.pool.take(shape, self.dev)
        self._recent_bufs.append(t)
        return t

    def _gnew(self, v):
        v.gpool = True
        return self.pool.take(v.t.shape, self.dev)

    def _tmp(self, *shape):
        self._gtmp += 1
        return self.bufs.get(f"gtmp.{self._gtmp % 3}.{'x'.join(map(str, shape))}", *shape, device=self.dev)

    def _tmp2(self, *shape):
        return self.bufs.get(f"gtmp2.{'x'.join(map(str, shape))}", *shape, device=self.dev)

    def var(self, t, need=True):
        self.n += 1
        v = Var(t, need, self.n)
        if self.save:           # (an evaluation tape must not hold references: torch's allocator frees what has no reader left)
            self._recent_vars.append(v)
        return v

    def _out(self, t, need=True):
        v = Var(t, need, self.n)
        if self.save:
            self._recent_vars.append(v)
        return v

    def acc(self, v, producer):
        """Add a gradient contribution to v: producer(out) writes it into a contiguous tensor of v's shape."""
        if not v.need:
            return
        if v.g is None:
            v.g = self._gnew(v)
            producer(v.g)
        else:
            tmp = self._tmp(*v.t.shape)
            producer(tmp)
            ops.axpby(v.g, tmp, 1.0, 1.0)

    # ---- parameter gradients (weight sharing: the first use overwrites, later uses add)
    def begin_backward(self, grads):
        self.grads, self._written = grads, set()

    def gparam(self, name, producer, shape=None, rows=None):
        """rows = (lo, hi): the producer fills rows lo .. hi - 1 of the parameter's gradient (a conv run as slices of its
        output channels: ENLCN's F -> 4F upsampler convs)."""
        gt = self.grads[name]
        key = name
        if rows is not None:
            gt = gt[rows[0]:rows[1]]
            key = (name, rows[0])
            self._written.add(name)
        if key not in self._written:
            self._written.add(key)
            producer(gt)
        else:
            tmp = self._tmp(*gt.shape)
            producer(tmp)
            ops.axpby(gt, tmp, 1.0, 1.0)

    # ---- ops
    def conv(self, x, key, names, act=None, prelu=None, add=None, relu=False, res=None, gelu=False):
        """x -> conv (bank entry `key`) [-> PixelShuffle / after PixelUnshuffle for the strided forms].
        names = (weight parameter name, bias parameter name or None).
        relu / res = (Var, factor) (plain 3x3 convs): out = relu(conv(x)) / out = Var + factor * conv(x) as the conv's
        epilogue in both modes (the ResBlock of the EDSR-body nets, network_nlsn.py:72-98: two launches instead of six); the
        backward masks / scales the incoming gradient in place before the conv's own.
        gelu: nn.GELU() behind the conv -- its epilogue in evaluation mode (epi 11), the separate op with the tape recording
        (the backward needs the pre-activation).
        prelu = (slope parameter, its name), add = (Var, factor): out = prelu(conv(x)) + factor * Var (DBPN's projection
        units).  In evaluation mode both ride in the 3x3 conv's epilogue (srhip_conv3x3_nhwc_split_ex epi 9 / 10; the
        addend of a transposed conv is added behind its PixelShuffle); with the tape recording they are the ordinary ops."""
        e = self.bank.d[key]
        if e.kind == "down":
            x = self.unshuffle(x, e.s)
        B, H, W, Ci = x.t.shape
        assert Ci == e.Ci, (key, x.t.shape, e.Ci)
        T = B * H * W
        y = self.new(B, H, W, e.Co)
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        cks = _chunks(B, H * W * max(Ci, e.Co))
        fuse = prelu is not None and not self.save and e.kind != "c1" and e.use_planes
        fuse_add = fuse and add is not None and e.kind != "deconv"
        radd = None
        if fuse_add:
            radd = add[0].t if add[0].t.is_contiguous() else add[0].t.contiguous()
        assert not (relu or res is not None) or (e.kind == "c3" and prelu is None and add is None and act is None and not (relu and res))
        rres = None
        if res is not None:
            rres = res[0].t if res[0].t.is_contiguous() else res[0].t.contiguous()
        for b0, b1 in cks:
            if e.kind == "c1":
                ops.gemm_nt(xin[b0:b1].view(-1, Ci), e.w1, e.bias, out=y[b0:b1].view(-1, e.Co))
            elif relu:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1], epi=1)
            elif gelu and not self.save and e.kind == "c3" and e.use_planes and e.wp.fmt == 1 and e.Co % 64 == 0:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1], epi=11)
            elif res is not None:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1], epi=2, R=rres[b0:b1], alpha=float(res[1]))
            elif fuse_add:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1], epi=10, R=radd[b0:b1], alpha=float(add[1]),
                            slope=prelu[0].data)
            elif fuse:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1], epi=9, slope=prelu[0].data)
            else:
                ops.conv3x3(xin[b0:b1], e.wp, e.bias, e.Co, out=y[b0:b1])
        out = self._out(y)
        if self.save:
            wname, bname = names[0], names[1]
            rows = names[2] if len(names) > 2 else None          # this conv = output channels rows[0] .. rows[1] - 1 of the parameter

            def bwd(x=x, out=out, e=e, wname=wname, bname=bname, rows=rows):
                g = out.g
                if g is None:
                    return
                if relu:                                     # out = relu(conv): the gradient where the output is positive
                    ops.relu_mask(g, out.t)
                if res is not None:                          # out = R + f conv: R takes g, the conv f g
                    self.acc(res[0], lambda o: o.copy_(g))
                    if float(res[1]) != 1.0:
                        ops.axpby(g, g, float(res[1]), 0.0)
                B, H, W, Ci = x.t.shape
                xin = x.t if x.t.is_contiguous() else x.t.contiguous()
                cks = _chunks(B, H * W * max(Ci, e.Co))
                wshape = (e.Co, e.Ci) if e.kind == "c1" else (e.Co, e.Ci, 3, 3)
                dW, db = self._tmp(*wshape), self._tmp(e.Co)
                for i, (b0, b1) in enumerate(cks):          # weight gradient: summed over the batch slices
                    dWi, dbi = (dW, db) if i == 0 else (self._tmp2(*wshape), self._tmp2(e.Co))
                    if e.kind == "c1":
                        ops.linear_wgrad(g[b0:b1].view(-1, e.Co), xin[b0:b1].view(-1, Ci), dWi, dbi)
                    else:
                        ops.conv3x3_wgrad(g[b0:b1], xin[b0:b1], dWi, dbi)
                    if i:
                        ops.axpby(dW, dWi, 1.0, 1.0)
                        ops.axpby(db, dbi, 1.0, 1.0)
                if e.kind == "c1":
                    self.gparam(wname, lambda o, dW=dW: o.view(e.Co, e.Ci).copy_(dW))
                    if bname:
                        self.gparam(bname, lambda o, db=db: o.copy_(db))

                    def dgrad1(o):
                        for b0, b1 in cks:
                            ops.gemm_nt(g[b0:b1].view(-1, e.Co), e.w1T, None, out=o[b0:b1].view(-1, Ci))
                    self.acc(x, dgrad1)
                    return
                if e.kind == "c3":
                    self.gparam(wname, lambda o, dW=dW: o.copy_(dW), rows=rows)
                    if bname:
                        self.gparam(bname, lambda o, db=db: o.copy_(db), rows=rows)
                elif e.kind == "deconv":
                    Co = e.Co // (e.s * e.s)
                    self.gparam(wname, lambda o, dW=dW: o.copy_(collapse_deconv(dW, e.Ci, Co, e.s, e.k, e.p)))
                    if bname:
                        self.gparam(bname, lambda o, db=db: o.copy_(db.view(Co, e.s * e.s).sum(1)))
                else:
                    Cin = e.Ci // (e.s * e.s)
                    self.gparam(wname, lambda o, dW=dW: o.copy_(collapse_down(dW, e.Co, Cin, e.s, e.k, e.p)))
                    if bname:
                        self.gparam(bname, lambda o, db=db: o.copy_(db))

                def dgrad3(o):
                    for b0, b1 in cks:
                        ops.conv3x3(g[b0:b1], e.wpt, None, Ci, out=o[b0:b1])
                self.acc(x, dgrad3)
            self.back.append(bwd)
        if e.kind == "deconv":
            if fuse and add is not None and ops.pixel_shuffle_add_ok(e.Co // (e.s * e.s), e.s):
                # evaluation: the addend of a transposed conv goes in with its PixelShuffle
                yh = self.new(B, H * e.s, W * e.s, e.Co // (e.s * e.s))
                ra = add[0].t if add[0].t.is_contiguous() else add[0].t.contiguous()
                ops.pixel_shuffle(out.t, e.s, nhwc_out=True, out=yh, add=ra, fac=float(add[1]))
                return self._out(yh)
            out = self.shuffle(out, e.s)
        if act is not None:
            out = act(out)
        if gelu and not (not self.save and e.kind == "c3" and e.use_planes and e.wp.fmt == 1 and e.Co % 64 == 0):
            out = self.unary(out, "gelu")
        if prelu is not None and not fuse:
            out = self.prelu(out, prelu[0], prelu[1])
        if add is not None and not fuse_add:
            out = self.axpby(out, add[0], 1.0, float(add[1]))
        return out

    def enlca(self, x, keys, proj, res_scale):
        """ENLCA (network_enlcn.py:330-366): keys = bank entries (= parameter name prefixes) of conv_match1 / conv_match2 /
        conv_assembly (1x1), proj = the stored projection matrix [F, d].  Linear attention as two dense products per
        sample, the normaliser carried as one more column of the value matrix.  Training: the intermediates stay (embeddings,
        normalisation factors, feature maps, per-sample context and numerator) and the backward runs the same products
        transposed around three small elementwise kernels (srhip_enlca_finish_bwd, srhip_performer_features_bwd,
        srhip_l2norm_rows_bwd); the contrastive term the reference computes in training mode is dropped by it (:434-437)."""
        e1, e2, e3 = (self.bank.d[k] for k in keys)
        B, H, W, C = x.t.shape
        T, L, d, Cy, Fn = B * H * W, H * W, e1.Co, e3.Co, proj.shape[0]
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        x2 = xin.view(T, C)
        dev = self.dev
        q, k = torch.empty(T, d, device=dev), torch.empty(T, d, device=dev)
        vext = torch.empty(T, Cy + 4, device=dev)
        vext[:, Cy] = 1.0
        vext[:, Cy + 1:] = 0.0
        ops.gemm_nt(x2, e1.w1, e1.bias, out=q)
        ops.gemm_nt(x2, e2.w1, e2.bias, out=k)
        ops.gemm_nt(x2, e3.w1, e3.bias, out=vext[:, :Cy])
        kk = 6.0 ** 0.5
        facq = fack = None
        if self.save:
            facq, fack = torch.empty(T, device=dev), torch.empty(T, device=dev)
            ops.l2norm_rows_train_(q, facq, kk)
            ops.l2norm_rows_train_(k, fack, kk)
        else:
            ops.l2norm_rows_(q, kk)
            ops.l2norm_rows_(k, kk)
        pj = proj.contiguous()
        fq = ops.performer_features_(ops.gemm_nt(q, pj), q)
        fk = ops.performer_features_(ops.gemm_nt(k, pj), k)
        y = self.new(B, H, W, Cy)
        y2 = y.view(T, Cy)
        colsum = torch.empty(Cy + 4, device=dev)
        # inference: one context / numerator buffer serves every sample; training keeps them per sample
        ctx_all = torch.empty(B if self.save else 1, Cy + 4, Fn, device=dev)
        num_all = torch.empty(T if self.save else L, Cy + 4, device=dev)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            ctx_t = ctx_all[b if self.save else 0]
            num = num_all[rows] if self.save else num_all
            ops.linear_wgrad(vext[rows], fk[rows], ctx_t, colsum)          # [v | 1]^T k' = (context | sum k')^T
            ops.gemm_nt(fq[rows], ctx_t, None, out=num)
            ops.enlca_finish(num, x2[rows], y2[rows], res_scale)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                g = out.g
                if g is None:
                    return
                g2 = (g if g.is_contiguous() else g.contiguous()).view(T, Cy)
                dnum = torch.empty(T, Cy + 4, device=dev)
                ops.enlca_finish_bwd(g2, num_all, dnum, res_scale)
                dfq, dfk = torch.empty(T, Fn, device=dev), torch.empty(T, Fn, device=dev)
                dvext = torch.empty(T, Cy + 4, device=dev)
                dctx_t, junk = torch.empty(Cy + 4, Fn, device=dev), torch.empty(Cy + 4, device=dev)
                for b in range(B):
                    rows = slice(b * L, (b + 1) * L)
                    ctx_t = ctx_all[b]
                    # num = fq ctx_t^T, ctx_t = vext^T fk
                    ops.gemm_nt(dnum[rows], ctx_t.t().contiguous(), None, out=dfq[rows])          # dfq = dnum ctx_t
                    ops.linear_wgrad(dnum[rows], fq[rows], dctx_t, junk)                          # dctx_t = dnum^T fq
                    ops.gemm_nt(fk[rows], dctx_t, None, out=dvext[rows])                          # dvext = fk dctx_t^T
                    ops.gemm_nt(vext[rows], dctx_t.t().contiguous(), None, out=dfk[rows])         # dfk = vext dctx_t
                ops.performer_features_bwd_(dfq, fq)                                              # -> d dash
                ops.performer_features_bwd_(dfk, fk)
                pjT = pj.t().contiguous()
                dq, dk = ops.gemm_nt(dfq, pjT), ops.gemm_nt(dfk, pjT)                             # dash = data P^T
                ops.l2norm_rows_bwd_(dq, q, facq, kk)
                ops.l2norm_rows_bwd_(dk, k, fack, kk)
                dv = dvext[:, :Cy].contiguous()
                for e, dY, key in ((e1, dq, keys[0]), (e2, dk, keys[1]), (e3, dv, keys[2])):
                    dW, db = torch.empty(e.Co, e.Ci, device=dev), torch.empty(e.Co, device=dev)
                    ops.linear_wgrad(dY, x2, dW, db)
                    self.gparam(key + ".0.weight", lambda o, dW=dW, e=e: o.view(e.Co, e.Ci).copy_(dW))
                    self.gparam(key + ".0.bias", lambda o, db=db: o.copy_(db))

                def dgrad(o):                       # d x = dq0 Wq + dk0 Wk + dv Wv + the residual's dout
                    o2 = o.view(T, C)
                    tmp = torch.empty(T, C, device=dev)
                    ops.gemm_nt(dq, e1.w1T, None, out=o2)
                    ops.gemm_nt(dk, e2.w1T, None, out=tmp)
                    ops.axpby(o2, tmp, 1.0, 1.0)
                    ops.gemm_nt(dv, e3.w1T, None, out=tmp)
                    ops.axpby(o2, tmp, 1.0, 1.0)
                    ops.axpby(o2, g2, 1.0, 1.0)
                self.acc(x, dgrad)
            self.back.append(bwd)
        return out

    def nlsa(self, x, keys, n_hashes, chunk_size, res_scale, rotations=None, tap=None, order=None):
        """NonLocalSparseAttention (network_nlsn.py:131-268): keys = bank entries (= parameter name prefixes) of conv_match
        (3x3, C -> C/4) and conv_assembly (1x1).  rotations: the LSH rotations [1, C/4, n_hashes, hash_buckets // 2] (None:
        drawn with torch.randn on the device, as the reference does at every call); tap: dict that receives the rotations
        and the token order used; order (tests): int64 [B, n_hashes, L], token index in the low 20 bits, used instead of the
        sort's result -- replays the order another implementation's sort produced, ties included.
        Forward: the fused kernels of nlsa.hip (hash, one radix sort, chunk attention written to token positions, rounds
        combined).  Backward (training): the embeddings are ordinary tape convs; the attention core is differentiated in
        chunk-major dense form -- rows gathered by the saved order (hash codes are constants of the step, as .detach() makes
        them in the reference, :171), P = softmax(xb xm^T) recomputed, then dP = dret yb^T, d logits = P (dP - sum P dP +
        d lse), dxb, dxm (through the keys' L2 normalisation), dyb as batched exact-f32 GEMMs (srhip_gemm_nt_batched) and
        scattered back to the tokens (padded rows are duplicates whose outputs the forward drops: zero gradient in)."""
        e1, e2 = (self.bank.d[k] for k in keys)
        B, H, W, C = x.t.shape
        L, T = H * W, B * H * W
        if self.save:
            xe_v = self.conv(x, keys[0], (keys[0] + ".0.weight", keys[0] + ".0.bias"))
            ye_v = self.conv(x, keys[1], (keys[1] + ".0.weight", keys[1] + ".0.bias"))
            xe, ye = xe_v.t, ye_v.t.view(T, e2.Co)
            xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        else:
            xin = x.t if x.t.is_contiguous() else x.t.contiguous()
            xe = torch.empty(B, H, W, e1.Co, device=self.dev)
            ops.conv3x3(xin, e1.wp, e1.bias, e1.Co, out=xe)
            ye = torch.empty(T, e2.Co, device=self.dev)
            ops.gemm_nt(xin.view(T, C), e2.w1, e2.bias, out=ye)
        hb = ops.nlsa_hash_buckets(L, chunk_size)
        if rotations is None:
            rotations = torch.randn(1, e1.Co, n_hashes, hb // 2, device=self.dev)
        assert tuple(rotations.shape) == (1, e1.Co, n_hashes, hb // 2), (rotations.shape, hb)
        if order is None:
            order = ops.nlsa_order(xe.view(T, e1.Co), rotations, B, L)
        else:
            assert tuple(order.shape) == (B, n_hashes, L) and order.dtype == torch.int64 and order.is_contiguous()
        if tap is not None:
            tap["rotations"], tap["order"] = rotations, order
        y = self.new(B, H, W, C)
        if not self.save:
            ops.nlsa_attention(xe.view(T, e1.Co), ye, order, xin.view(T, C), y.view(T, C), B, L, chunk_size, res_scale)
            return self._out(y)
        _, ret, score = ops.nlsa_attention(xe.view(T, e1.Co), ye, order, xin.view(T, C), y.view(T, C), B, L, chunk_size,
                                           res_scale, keep=True)
        out = self._out(y)
        dev, cs, nh, Ce, Cy = self.dev, int(chunk_size), int(n_hashes), e1.Co, e2.Co

        def bwd(x=x, out=out, xe_v=xe_v, ye_v=ye_v):
            g = out.g
            if g is None:
                return
            g2 = (g if g.is_contiguous() else g.contiguous()).view(B, L, Cy)
            # ---- the rounds' softmax combination (:258-262), token order
            probs = torch.softmax(score, dim=1)                                              # [B, nh, L]
            gx = g2.view(B, 1, L, Cy).expand(B, nh, L, Cy).contiguous()
            a = ops.rowdot(ret.reshape(-1, Cy), gx.view(-1, Cy)).view(B, nh, L) * res_scale  # d out / d probs
            dlse = probs * (a - (probs * a).sum(1, keepdim=True))
            dret = gx.mul_((probs * res_scale).unsqueeze(-1))                                # [B, nh, L, Cy]
            # ---- chunk-major index maps from the saved order (positions past L repeat the last `padding` positions, :213-218)
            tok = order & ((1 << 20) - 1)                                                    # [B, nh, L] int64
            padding = (cs - L % cs) % cs
            tokp = torch.cat([tok, tok[:, :, L - padding:]], 2) if padding else tok
            Lp = L + padding
            nch = Lp // cs
            NB = B * nh * nch
            qtok = tokp.view(B, nh, nch, cs)
            ktok = torch.cat([qtok, qtok.roll(1, 2), qtok.roll(-1, 2)], 3)                   # own, previous, next chunk (:224-231)
            base = (torch.arange(B, device=dev) * L).view(B, 1, 1, 1)
            qflat, kflat = (qtok + base).reshape(-1), (ktok + base).reshape(-1)
            rbase = (torch.arange(B * nh, device=dev) * L).view(B, nh, 1, 1)
            rflat = (qtok + rbase).reshape(-1)                                               # row of (sample, round, token)
            xe2, ye2 = xe.view(T, Ce), ye
            xb = xe2.index_select(0, qflat)                                                  # [NB cs, Ce] queries (un-normalised)
            xm = xe2.index_select(0, kflat)                                                  # [NB 3cs, Ce] keys
            fac = torch.empty(xm.shape[0], device=dev)
            ops.l2norm_rows_train_(xm, fac, 1.0)                                             # F.normalize(eps 5e-5) (:222)
            yb = ye2.index_select(0, kflat)                                                  # [NB 3cs, Cy] values
            dret_s = dret.view(-1, Cy).index_select(0, rflat)
            dlse_s = dlse.reshape(-1).index_select(0, rflat)
            if padding:                                                                      # padded rows: outputs dropped (:246-249)
                dret_s.view(B, nh, Lp, Cy)[:, :, L:] = 0.0
                dlse_s.view(B, nh, Lp)[:, :, L:] = 0.0
            del dret, gx
            K3 = 3 * cs
            P = torch.empty(NB * cs, K3, device=dev)
            ops.gemm_nt_batched(xb[:cs], (cs * Ce, 0), xm[:K3], (K3 * Ce, 0), P[:cs], (cs * K3, 0), cs, K3, Ce, NB, 1)
            junk = torch.empty(NB * cs, device=dev)
            ops.softmax_rows_lse_(P, junk)                                                   # P = exp(raw - lse) (:236-240)
            dP = torch.empty(NB * cs, K3, device=dev)
            ops.gemm_nt_batched(dret_s[:cs], (cs * Cy, 0), yb[:K3], (K3 * Cy, 0), dP[:cs], (cs * K3, 0), cs, K3, Cy, NB, 1)
            ops.softmax_rows_bwd_(P, dP, dlse_s)                                             # -> d raw
            t3 = lambda m, r, c_: m.view(NB, r, c_).transpose(1, 2).contiguous()
            xmT, xbT, drawT, PT, dretT = t3(xm, K3, Ce), t3(xb, cs, Ce), t3(dP, cs, K3), t3(P, cs, K3), t3(dret_s, cs, Cy)
            dxb = torch.empty(NB * cs, Ce, device=dev)
            ops.gemm_nt_batched(dP[:cs], (cs * K3, 0), xmT.view(-1, K3)[:Ce], (Ce * K3, 0), dxb[:cs], (cs * Ce, 0), cs, Ce, K3, NB, 1)
            dxm = torch.empty(NB * K3, Ce, device=dev)
            ops.gemm_nt_batched(drawT.view(-1, cs)[:K3], (K3 * cs, 0), xbT.view(-1, cs)[:Ce], (Ce * cs, 0), dxm[:K3], (K3 * Ce, 0),
                                K3, Ce, cs, NB, 1)
            dyb = torch.empty(NB * K3, Cy, device=dev)
            ops.gemm_nt_batched(PT.view(-1, cs)[:K3], (K3 * cs, 0), dretT.view(-1, cs)[:Cy], (Cy * cs, 0), dyb[:K3], (K3 * Cy, 0),
                                K3, Cy, cs, NB, 1)
            ops.l2norm_rows_bwd_(dxm, xm, fac, 1.0)                                          # keys' normalisation
            dxe = torch.zeros(T, Ce, device=dev)
            dxe.index_add_(0, qflat, dxb)
            dxe.index_add_(0, kflat, dxm)
            dye = torch.zeros(T, Cy, device=dev)
            dye.index_add_(0, kflat, dyb)
            self.acc(xe_v, lambda o: o.view(T, Ce).copy_(dxe))
            self.acc(ye_v, lambda o: o.view(T, Cy).copy_(dye))
            self.acc(x, lambda o: o.view(B, L, Cy).copy_(g2))                                # the residual (:266)
        self.back.append(bwd)
        return out

    def _no_backward(self, what):
        if self.save:
            def bwd():
                raise NotImplementedError(f"{what} on libsrhip: inference only; no backward")
            self.back.append(bwd)

    def unary(self, x, kind):
        """nn.GELU() / sigmoid as their own op (DFCAN: network_dfcan.py:44-47,98-99,108,111-113)."""
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        y = self.new(*x.t.shape)
        ops.unary(xin, y, kind)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin, y=y):
                if out.g is None:
                    return
                g = out.g if out.g.is_contiguous() else out.g.contiguous()
                self.acc(x, lambda o: ops.unary_bwd(xin if kind == "gelu" else y, g, o, kind))
            self.back.append(bwd)
        return out

    def fourier_gate(self, x0, x1, key_conv, names_conv, w1, b1, w2, b2, names_gate=None, gamma=0.8, eps=1e-8):
        """DFCAN's Fourier channel attention (RCAB.forward, network_dfcan.py:60-70): gate = sigmoid(W2 relu(W1 avgpool(relu(
        conv(fftshift(|FFT2(x1)|^0.8)))))), out = x0 + x1 * gate.  names_gate = parameter names of (W1, b1, W2, b2).
        Backward (training): the gate's two tiny Linears by hand on [B, C] tensors (ops.mm: the library's exact-f32 GEMM); the
        spectrum magnitude through the in-tree DFT passes (srhip_fft2_mag_pow_shift_bwd) -- with F = FFT2(x1) and G the
        incoming gradient un-shifted and times gamma (|F| + eps)^(gamma - 1) / |F|, d x1 = Re(unnormalised IFFT2(G F)) --
        registered in front of the conv so that it runs behind the conv's backward."""
        a = x1.t if x1.t.is_contiguous() else x1.t.contiguous()
        B, H, W, C = a.shape
        m = torch.empty_like(a)
        ops.fft2_mag_pow_shift(a, m, gamma, eps)
        mv = self.var(m, need=self.save)
        if self.save:
            def spec_bwd(x1=x1, mv=mv, a=a):
                if mv.g is None:
                    return
                gm = mv.g if mv.g.is_contiguous() else mv.g.contiguous()
                self.acc(x1, lambda o: ops.fft2_mag_pow_shift_bwd(a, gm, o, gamma, eps))
            self.back.append(spec_bwd)
        c = self.relu(self.conv(mv, key_conv, names_conv))
        y = self.new(*a.shape)
        x0c = x0.t if x0.t.is_contiguous() else x0.t.contiguous()
        ops.channel_gate(c.t, w1, b1, w2, b2, x0c, a, y)
        out = self._out(y)
        if self.save:
            assert names_gate is not None
            gate = ops.SCRATCH.get("gate_vec", B * C, device=a.device)[:B * C].view(B, C).clone()
            pool = c.t.mean((1, 2))

            def gate_bwd(x0=x0, x1=x1, c=c, out=out, a=a, gate=gate, pool=pool):
                g = out.g
                if g is None:
                    return
                self.acc(x0, lambda o: o.copy_(g))
                self.acc(x1, lambda o: torch.mul(g, gate.view(B, 1, 1, C), out=o))
                dgate = (g * a).sum((1, 2))
                z1 = ops.mm(pool, w1, tb=True, bias=b1)
                r1 = torch.relu(z1)
                dz2 = dgate * gate * (1.0 - gate)
                dz1 = ops.mm(dz2, w2) * (z1 > 0)
                self.gparam(names_gate[2], lambda o: o.view(w2.shape).copy_(ops.mm(dz2, r1, ta=True)))
                self.gparam(names_gate[3], lambda o: o.copy_(dz2.sum(0)))
                self.gparam(names_gate[0], lambda o: o.view(w1.shape).copy_(ops.mm(dz1, pool, ta=True)))
                self.gparam(names_gate[1], lambda o: o.copy_(dz1.sum(0)))
                dpool = ops.mm(dz1, w1) / float(H * W)
                self.acc(c, lambda o: o.copy_(dpool.view(B, 1, 1, C).expand(B, H, W, C)))
            self.back.append(gate_bwd)
        return out

    # ---- ops on token matrices and their maps to / from feature maps (ACT's transformer branch in training)
    @staticmethod
    def _c(t):
        return t if t.is_contiguous() else t.contiguous()

    def linear(self, x, weight, bias, wname, bname=None):
        """nn.Linear / a 1x1 conv on rows: x [M, K] -> [M, N] on the exact-f32 GEMM; weight [N, K(, 1, 1)]."""
        w = weight.data.reshape(weight.shape[0], -1)
        N, K = w.shape
        xin = self._c(x.t)
        M = xin.shape[0]
        assert xin.shape == (M, K), (wname, xin.shape, w.shape)
        y = self.new(M, N)
        ops.gemm_nt(xin, w, None if bias is None else bias.data, out=y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin, w=w):
                if out.g is None:
                    return
                g = self._c(out.g)
                N4 = (N + 3) & ~3
                if N4 != N:                  # the contraction of the data gradient runs over N: zero columns up to a multiple of 4
                    g = torch.nn.functional.pad(g, (0, N4 - N))
                if wname is not None:        # (None: a constant matrix, e.g. the averaging weights of a pooling)
                    dW, db = self._tmp(N4, K), self._tmp(N4)
                    ops.linear_wgrad(g, xin, dW, db)
                    self.gparam(wname, lambda o: o.view(N, K).copy_(dW[:N]))
                    if bname:
                        self.gparam(bname, lambda o: o.copy_(db[:N]))
                if x.need:
                    wT = torch.nn.functional.pad(w.t(), (0, N4 - N)).contiguous()
                    self.acc(x, lambda o: ops.gemm_nt(g, wT, None, out=o))
            self.back.append(bwd)
        return out

    def layernorm_rows(self, x, ln, gname, bname, eps=None):
        """nn.LayerNorm over the rows of x [M, C] (any C up to 2048); eps: the module's unless given."""
        xin = self._c(x.t)
        y = self.new(*xin.shape)
        eps = float(ln.eps if eps is None else eps)
        ops.layernorm_rows(xin, ln.weight.data, ln.bias.data, y, eps=eps)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is None:
                    return
                C = xin.shape[1]
                dx, dg, db = self._tmp(*xin.shape), self._tmp(C), self._tmp2(C)
                ops.layernorm_rows_bwd(self._c(out.g), xin, ln.weight.data, dx, dg, db, eps=eps)
                self.gparam(gname, lambda o: o.copy_(dg))
                self.gparam(bname, lambda o: o.copy_(db))
                self.acc(x, lambda o: o.copy_(dx))
            self.back.append(bwd)
        return out

    def cols(self, x, lo, hi):
        """x[..., lo:hi] as its own tensor (torch.split / chunk on the last dim): a copy; the adjoint places the gradient."""
        y = self.new(*x.t.shape[:-1], hi - lo)
        y.copy_(x.t[..., lo:hi])
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is None:
                    return

                def prod(o):
                    o.zero_()
                    o[..., lo:hi].copy_(out.g)
                self.acc(x, prod)
            self.back.append(bwd)
        return out

    def cat_cols(self, vs):
        """torch.cat(vs, dim=-1) of tensors of any rank: copies; the adjoint slices."""
        Cs = [v.t.shape[-1] for v in vs]
        z = self.new(*vs[0].t.shape[:-1], sum(Cs))
        o = 0
        for v, c in zip(vs, Cs):
            z[..., o:o + c].copy_(v.t)
            o += c
        out = self._out(z)
        if self.save:
            def bwd(vs=vs, out=out):
                if out.g is None:
                    return
                o = 0
                for v, c in zip(vs, Cs):
                    self.acc(v, lambda dst, o=o, c=c: dst.copy_(out.g[..., o:o + c]))
                    o += c
            self.back.append(bwd)
        return out

    def unfold(self, x, k, s):
        """F.unfold(x, k, stride=s) of an NHWC map -> token rows [B * nT, C k k]; its adjoint is F.fold."""
        xin = self._c(x.t)
        B, H, W, C = xin.shape
        nT = ((H - k) // s + 1) * ((W - k) // s + 1)
        y = self.new(B * nT, C * k * k)
        ops.unfold(xin, C, k, s, 0, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: ops.fold(self._c(out.g), C, k, s, o))
            self.back.append(bwd)
        return out

    def fold(self, tok, C, k, s, B, H, W):
        """F.fold(tok, (H, W), k, stride=s) -> NHWC [B, H, W, C] (overlaps add); its adjoint is F.unfold."""
        tin = self._c(tok.t)
        y = self.new(B, H, W, C)
        ops.fold(tin, C, k, s, y)
        out = self._out(y)
        if self.save:
            def bwd(tok=tok, out=out):
                if out.g is not None:
                    self.acc(tok, lambda o: ops.unfold(self._c(out.g), C, k, s, 0, o))
            self.back.append(bwd)
        return out

    def conv_im2col(self, x, weight, bias, wname, bname, k, relu=False):
        """k x k conv (padding k // 2) as im2col + GEMM (ACT's 5 x 5 head convs, network_act.py:362-364; GRL's C/4-channel
        convs).  Backward: the weight gradient from the re-made patch matrix, the data gradient as the same conv of the
        gradient with the flipped, transposed kernel.  Contraction lengths that are not multiples of 4 are zero-padded."""
        xin = self._c(x.t)
        B, H, W, C = xin.shape
        Co = weight.shape[0]
        T = B * H * W
        K, Kg = C * k * k, Co * k * k
        K4, Kg4 = (K + 3) & ~3, (Kg + 3) & ~3
        assert T * max(K4, Kg4) < (1 << 29), "conv_im2col: the patch matrix of this batch passes 2 GiB"

        def patches(src, Cs, Kp):
            cb = self._tmp2(T, Kp)
            if Kp != Cs * k * k:
                cb.zero_()
            ops.unfold(src, Cs, k, 1, k // 2, cb)
            return cb

        def padded(w2, Kp):
            return w2 if w2.shape[1] == Kp else torch.nn.functional.pad(w2, (0, Kp - w2.shape[1])).contiguous()
        y = self.new(B, H, W, Co)
        ops.gemm_nt(patches(xin, C, K4), padded(weight.data.reshape(Co, K), K4), bias.data, out=y.view(T, Co))
        if relu:
            ops.leaky_relu_(y, 0.0)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin, y=y):
                if out.g is None:
                    return
                g = self._c(out.g)
                if relu:
                    ops.relu_mask(g, y)
                Co4 = (Co + 3) & ~3
                g2 = g.view(T, Co) if Co4 == Co else torch.nn.functional.pad(g.view(T, Co), (0, Co4 - Co))
                dW, db = self._tmp(Co4, K4), self._tmp(Co4)
                ops.linear_wgrad(g2, patches(xin, C, K4), dW, db)
                self.gparam(wname, lambda o: o.view(Co, K).copy_(dW[:Co, :K]))
                self.gparam(bname, lambda o: o.copy_(db[:Co]))
                if x.need:
                    wt = padded(weight.data.flip(2, 3).permute(1, 0, 2, 3).reshape(C, Kg), Kg4)
                    gcols = patches(g, Co, Kg4)
                    self.acc(x, lambda o: ops.gemm_nt(gcols, wt, None, out=o.view(T, C)))
            self.back.append(bwd)
        return out

    def attend(self, q, k, v, B, Tq, Tk, heads, dh, scale, bias=None, on_dbias=None):
        """softmax(scale q k^T) v per (sample, head): q [B Tq, heads dh], k / v [B Tk, heads dh] -> [B Tq, heads dh]; the
        (sample, head) products as batched launches of the exact-f32 GEMM around the row softmax, operands re-laid per head
        (copies), contraction lengths zero-padded to multiples of 4.  Backward: dP = dO v^T, the softmax's row gradient,
        dq = dS k, dk = dS^T q, dv = P^T dO -- four more batched launches on transposed copies.
        bias [R, Tq, Tk], R a multiple of heads dividing B heads (heads: one image per sample, OmniSR's relative-position bias;
        nW heads with heads = 1 rows per head: GRL's bias + shift mask per window): added to the scaled logits periodically;
        on_dbias(d [R, Tq, Tk]) receives its gradient (the sum over the periods)."""
        Z = B * heads
        Tk4, Tq4 = (Tk + 3) & ~3, (Tq + 3) & ~3
        assert Z * Tq * Tk4 < (1 << 29), "attend: the attention matrices of this batch pass 2 GiB"
        dev = self.dev

        def heads_of(t, T):                                   # [B T, heads dh] -> [Z, T, dh]
            return t.reshape(B, T, heads, dh).permute(0, 2, 1, 3).reshape(Z, T, dh).contiguous()

        def padT(t, T, T4):                                   # [Z, T, n] -> [Z, n, T4], zero-padded transposes
            o = torch.zeros(Z, t.shape[2], T4, device=dev)
            o[:, :, :T].copy_(t.transpose(1, 2))
            return o

        def bmm(A, W, C, M, N, K):                            # C_z[M, N] = A_z[M, K] W_z[N, K]^T, contiguous [Z, ., .] views
            ops.gemm_nt_batched(A[0], (A.stride(0), 0), W[0], (W.stride(0), 0), C[0], (C.stride(0), 0), M, N, K, Z, 1)

        qh, kh, vh = heads_of(q.t, Tq), heads_of(k.t, Tk), heads_of(v.t, Tk)
        P = (self.new(Z, Tq, Tk4) if self.save else torch.empty(Z, Tq, Tk4, device=dev))
        P.zero_()
        bmm(qh, kh, P[:, :, :Tk], Tq, Tk, dh)
        if bias is None:
            ops.softmax_rows_(P.view(Z * Tq, Tk4)[:, :Tk], scale)
        else:
            R = bias.shape[0]
            assert Z % R == 0, (Z, R)
            bpad = torch.zeros(R, Tq, Tk4, device=dev)
            bpad[:, :, :Tk].copy_(bias)
            ops.axpby(P, P, float(scale), 0.0)
            ops.add_periodic(P, bpad)
            ops.softmax_rows_(P.view(Z * Tq, Tk4)[:, :Tk], 1.0)
        oh = torch.empty(Z, Tq, dh, device=dev)
        bmm(P, padT(vh, Tk, Tk4), oh, Tq, dh, Tk4)
        y = self.new(B * Tq, heads * dh)
        y.view(B, Tq, heads, dh).copy_(oh.view(B, heads, Tq, dh).permute(0, 2, 1, 3))
        out = self._out(y)
        if self.save:
            def bwd(q=q, k=k, v=v, out=out, P=P):
                if out.g is None:
                    return
                qh, kh, vh = heads_of(q.t, Tq), heads_of(k.t, Tk), heads_of(v.t, Tk)     # re-laid again: not kept
                gh = heads_of(self._c(out.g), Tq)
                dS = torch.zeros(Z, Tq, Tk4, device=dev)
                bmm(gh, vh, dS[:, :, :Tk], Tq, Tk, dh)
                ops.softmax_rows_bwd_(P.view(Z * Tq, Tk4)[:, :Tk], dS.view(Z * Tq, Tk4)[:, :Tk])
                if bias is not None and on_dbias is not None:
                    db = torch.empty(bias.shape[0], Tq, Tk4, device=dev)
                    ops.sum_periodic(dS, db)
                    on_dbias(db[:, :, :Tk])
                ops.axpby(dS, dS, float(scale), 0.0)

                def back_to_rows(th, T):                      # [Z, T, dh] -> [B T, heads dh]
                    return lambda o: o.view(B, T, heads, dh).copy_(th.view(B, heads, T, dh).permute(0, 2, 1, 3))
                dq = torch.empty(Z, Tq, dh, device=dev)
                bmm(dS, padT(kh, Tk, Tk4), dq, Tq, dh, Tk4)
                self.acc(q, back_to_rows(dq, Tq))
                dST = padT(dS[:, :, :Tk], Tq, Tq4)            # [Z, Tk, Tq4]
                dk = torch.empty(Z, Tk, dh, device=dev)
                bmm(dST, padT(qh, Tq, Tq4), dk, Tk, dh, Tq4)
                self.acc(k, back_to_rows(dk, Tk))
                PT = padT(P[:, :, :Tk], Tq, Tq4)
                dv = torch.empty(Z, Tk, dh, device=dev)
                bmm(PT, padT(gh, Tq, Tq4), dv, Tk, dh, Tq4)
                self.acc(v, back_to_rows(dv, Tk))
            self.back.append(bwd)
        return out

    def rcan_gate(self, a, x, w1, b1, w2, b2, names):
        """RCAN's channel attention with the block's skip (network_act.py:230-277): out = x + a * sigmoid(W2 relu(W1 mean(a) + b1)
        + b2).  names = parameter names of (W1, b1, W2, b2); the two tiny Linears' backward by hand on [B, C] tensors."""
        ain, xin = self._c(a.t), self._c(x.t)
        B, H, W, C = ain.shape
        w1m, w2m = w1.data.reshape(w1.shape[0], -1).contiguous(), w2.data.reshape(w2.shape[0], -1).contiguous()
        y = self.new(B, H, W, C)
        ops.channel_gate(ain, w1m, b1.data, w2m, b2.data, xin, ain, y)
        out = self._out(y)
        if self.save:
            gate = ops.SCRATCH.get("gate_vec", B * C, device=ain.device)[:B * C].view(B, C).clone()
            pool = ain.mean((1, 2))

            def bwd(a=a, x=x, out=out, gate=gate, pool=pool):
                g = out.g
                if g is None:
                    return
                self.acc(x, lambda o: o.copy_(g))
                dgate = (g * ain).sum((1, 2))
                z1 = ops.mm(pool, w1m, tb=True, bias=b1.data)
                r1 = torch.relu(z1)
                dz2 = dgate * gate * (1.0 - gate)
                dz1 = ops.mm(dz2, w2m) * (z1 > 0)
                self.gparam(names[2], lambda o: o.view(w2m.shape).copy_(ops.mm(dz2, r1, ta=True)))
                self.gparam(names[3], lambda o: o.copy_(dz2.sum(0)))
                self.gparam(names[0], lambda o: o.view(w1m.shape).copy_(ops.mm(dz1, pool, ta=True)))
                self.gparam(names[1], lambda o: o.copy_(dz1.sum(0)))
                dpool = ops.mm(dz1, w1m) / float(H * W)
                self.acc(a, lambda o: torch.addcmul(dpool.view(B, 1, 1, C).expand(B, H, W, C), g, gate.view(B, 1, 1, C), out=o))
            self.back.append(bwd)
        return out

    # ---- more ops of this kind (OmniSR's training graph)
    def reshape(self, x, *shape):
        """the same storage under another shape (rows <-> NHWC map)"""
        r = self.var(self._c(x.t).view(*shape))
        if self.save:
            def bwd(x=x, r=r):
                if r.g is not None:
                    self.acc(x, lambda o: o.copy_(r.g.reshape(o.shape)))
            self.back.append(bwd)
        return r

    def relayout(self, x, fwd, inv, pad_last=None):
        """y = fwd(x) made dense, fwd a chain of view ops (reshape / permute) and inv its inverse: window / grid partitions.
        pad_last: the last dimension zero-padded to this width (a contraction length the GEMMs take); inv sees the unpadded part."""
        v = fwd(x.t)
        n = v.shape[-1]
        if pad_last is None or pad_last == n:
            y = self.new(*v.shape)
            y.copy_(v)
        else:
            y = self.new(*v.shape[:-1], pad_last)
            y.zero_()
            y[..., :n].copy_(v)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: o.copy_(inv(out.g[..., :n])))
            self.back.append(bwd)
        return out

    def dwconv(self, x, weight, bias, wname, bname=None):
        """nn.Conv2d(C, C, 3, padding=1, groups=C).  Backward: the data gradient is the same conv with flipped taps; the weight
        gradient dW[c][tap] = sum_p g[p][c] x[p + tap][c] is the block diagonal of g^T . unfold(x) (one GEMM; the other blocks
        are dropped)."""
        xin = self._c(x.t)
        B, H, W, C = xin.shape
        y = self.new(B, H, W, C)
        ops.dwconv3x3(xin, weight.data, None if bias is None else bias.data, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is None:
                    return
                g = self._c(out.g)
                T = B * H * W
                colsb = self._tmp2(T, 9 * C)
                ops.unfold(xin, C, 3, 1, 1, colsb)
                full, db = self._tmp(C, 9 * C), self._tmp(C)
                ops.linear_wgrad(g.view(T, C), colsb, full, db)
                ar = torch.arange(C, device=full.device)
                self.gparam(wname, lambda o: o.view(C, 9).copy_(full.view(C, C, 9)[ar, ar]))
                if bname:
                    self.gparam(bname, lambda o: o.copy_(db))
                wf = weight.data.flip(2, 3).contiguous()
                self.acc(x, lambda o: ops.dwconv3x3(g, wf, None, o))
            self.back.append(bwd)
        return out

    def se_gate(self, d, w1, w2, names):
        """squeeze-excitation of MBConv (network_omni_sr.py:133-148): out = d * sigmoid(W2 silu(W1 mean(d))), no biases; the two
        tiny Linears' backward by hand on [B, C] tensors."""
        din = self._c(d.t)
        B, H, W, C = din.shape
        w1m, w2m = w1.data, w2.data
        y = self.new(B, H, W, C)
        ops.channel_gate(din, w1m, None, w2m, None, None, din, y, mid_act="silu")
        out = self._out(y)
        if self.save:
            gate = ops.SCRATCH.get("gate_vec", B * C, device=din.device)[:B * C].view(B, C).clone()
            pool = din.mean((1, 2))

            def bwd(d=d, out=out, gate=gate, pool=pool):
                g = out.g
                if g is None:
                    return
                dgate = (g * din).sum((1, 2))
                z1 = ops.mm(pool, w1m, tb=True)
                sg = torch.sigmoid(z1)
                r1 = z1 * sg
                dz2 = dgate * gate * (1.0 - gate)
                dz1 = ops.mm(dz2, w2m) * (sg * (1.0 + z1 * (1.0 - sg)))
                self.gparam(names[1], lambda o: o.view(w2m.shape).copy_(ops.mm(dz2, r1, ta=True)))
                self.gparam(names[0], lambda o: o.view(w1m.shape).copy_(ops.mm(dz1, pool, ta=True)))
                dpool = ops.mm(dz1, w1m) / float(H * W)
                self.acc(d, lambda o: torch.addcmul(dpool.view(B, 1, 1, C).expand(B, H, W, C), g, gate.view(B, 1, 1, C), out=o))
            self.back.append(bwd)
        return out

    def gelu_gate(self, x):
        """gelu(x1) * x2 for the two channel halves of x [.., 2C] (Gated_Conv_FeedForward, network_omni_sr.py:324-327)."""
        xin = self._c(x.t)
        C2 = xin.shape[-1]
        C = C2 // 2
        T = xin.numel() // C2
        y = self.new(*xin.shape[:-1], C)
        ops.gelu_gate(xin.view(T, C2), y.view(T, C))
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is None:
                    return
                g = self._c(out.g).view(T, C)
                x1, x2 = xin.view(T, C2)[:, :C].contiguous(), xin.view(T, C2)[:, C:].contiguous()
                ga = ops.unary(x1, torch.empty_like(x1), "gelu")
                d2 = ops.mul(g, ga)
                d1 = ops.unary_bwd(x1, ops.mul(g, x2), torch.empty_like(x1), "gelu")

                def prod(o):
                    ov = o.view(T, C2)
                    ov[:, :C].copy_(d1)
                    ov[:, C:].copy_(d2)
                self.acc(x, prod)
            self.back.append(bwd)
        return out

    def mul_sigmoid(self, x, c):
        """x * sigmoid(c) (ESA, network_omni_sr.py:113-114)."""
        xin, cin = self._c(x.t), self._c(c.t)
        y = self.new(*xin.shape)
        ops.mul_sigmoid(xin, cin, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, c=c, out=out):
                if out.g is None:
                    return
                g = self._c(out.g)
                self.acc(x, lambda o: ops.mul_sigmoid(g, cin, o))
                sg = ops.unary(cin, torch.empty_like(cin), "sigmoid")
                self.acc(c, lambda o: ops.unary_bwd(sg, ops.mul(g, xin), o, "sigmoid"))
            self.back.append(bwd)
        return out

    def normalize_rows(self, x, eps=1e-12):
        """F.normalize(x, dim=-1) on rows [M, n]: x / max(|x|, eps); dx = (g - y (y . g)) / max(|x|, eps)."""
        xin = self._c(x.t)
        M, n = xin.shape
        inv = 1.0 / ops.rowdot(xin, xin).sqrt().clamp_min(eps)             # [M]: per-row scalars
        invE = inv.view(M, 1).expand(M, n).contiguous()
        y = self.new(M, n)
        ops.mul(xin, invE, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, y=y, invE=invE, inv=inv):
                if out.g is None:
                    return
                g = self._c(out.g)
                coef = (ops.rowdot(y, g) * inv).view(M, 1).expand(M, n).contiguous()

                def prod(o):
                    ops.mul(g, invE, o)
                    ops.axpby(o, ops.mul(y, coef), -1.0, 1.0)
                self.acc(x, prod)
            self.back.append(bwd)
        return out

    def scale_rows(self, x, s_rows, on_grad):
        """y[r] = s_rows[r] * x[r] (the head's temperature on the channel attention's query rows); on_grad receives
        d s_rows [M] = rowdot(g, x)."""
        xin = self._c(x.t)
        M, n = xin.shape
        sE = s_rows.view(M, 1).expand(M, n).contiguous()
        y = self.new(M, n)
        ops.mul(xin, sE, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, sE=sE):
                if out.g is None:
                    return
                g = self._c(out.g)
                on_grad(ops.rowdot(g, xin))
                self.acc(x, lambda o: ops.mul(g, sE, o))
            self.back.append(bwd)
        return out

    def conv_patches(self, x, weight, bias, wname, bname, k, s):
        """k x k conv, stride s, no padding, as F.unfold + GEMM (ESA's conv2, network_omni_sr.py:96,106).  Backward: weight
        gradient from the patch matrix, data gradient = F.fold of g W (the overlap-add adjoint)."""
        xin = self._c(x.t)
        B, H, W, C = xin.shape
        Co = weight.shape[0]
        Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
        T = B * Ho * Wo
        w2 = weight.data.reshape(Co, -1)
        colsb = self._tmp2(T, C * k * k)
        ops.unfold(xin, C, k, s, 0, colsb)
        y = self.new(B, Ho, Wo, Co)
        ops.gemm_nt(colsb, w2, bias.data, out=y.view(T, Co))
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is None:
                    return
                g = self._c(out.g).view(T, Co)
                colsb = self._tmp2(T, C * k * k)
                ops.unfold(xin, C, k, s, 0, colsb)
                dW, db = self._tmp(Co, C * k * k), self._tmp(Co)
                ops.linear_wgrad(g, colsb, dW, db)
                self.gparam(wname, lambda o: o.view(Co, C * k * k).copy_(dW))
                self.gparam(bname, lambda o: o.copy_(db))
                dcols = ops.gemm_nt(g, w2.t().contiguous(), None)
                self.acc(x, lambda o: ops.fold(dcols, C, k, s, o))
            self.back.append(bwd)
        return out

    def maxpool(self, x, k, s):
        xin = self._c(x.t)
        y = ops.maxpool2d(xin, k, s)
        self.n += 1
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is not None:
                    self.acc(x, lambda o: o.copy_(ops.maxpool2d_bwd(xin, self._c(out.g), k, s)))
            self.back.append(bwd)
        return out

    def bilinear(self, x, Ho, Wo):
        """F.interpolate(bilinear, align_corners=False) to (Ho, Wo).  The map is separable and linear: its adjoint runs as two
        GEMMs against the (tiny, zero-padded) interpolation matrices A_h [Ho, H], A_w [Wo, W]."""
        xin = self._c(x.t)
        B, H, W, C = xin.shape
        y = ops.bilinear_resize(xin, Ho, Wo)
        self.n += 1
        out = self._out(y)
        if self.save:
            def interp(n_in, n_out):                  # A^T zero-padded to a multiple of 4 rows: [n_in4, n_out]
                eye = torch.eye(n_in, device=xin.device).view(1, n_in, n_in, 1)
                a = torch.nn.functional.interpolate(eye, size=(n_out, 1), mode="bilinear", align_corners=False)[0, :, :, 0]
                at = torch.zeros((n_in + 3) & ~3, n_out, device=xin.device)
                at[:n_in].copy_(a)
                return at

            def bwd(x=x, out=out):
                if out.g is None:
                    return
                g = self._c(out.g)                                              # [B, Ho, Wo, C]
                ah, aw = interp(H, Ho), interp(W, Wo)
                r1 = g.permute(0, 2, 3, 1).reshape(B * Wo * C, Ho).contiguous()   # rows (b, x, c), columns y
                t1 = ops.gemm_nt(r1, ah, None)[:, :H]                             # [(b, x, c), i]
                r2 = t1.reshape(B, Wo, C, H).permute(0, 3, 2, 1).reshape(B * H * C, Wo).contiguous()   # rows (b, i, c), columns x
                t2 = ops.gemm_nt(r2, aw, None)[:, :W]                             # [(b, i, c), j]
                self.acc(x, lambda o: o.copy_(t2.reshape(B, H, C, W).permute(0, 1, 3, 2)))
            self.back.append(bwd)
        return out

    def window_heads(self, m, wsz, heads, shift=0, pad=None):
        """NHWC map [B, Hm, Wm, heads dh] -> rows [(b, window, head, token), dh (zero-padded to `pad`)]: roll by -shift,
        window partition, one row block per head.  Returns (Var, fwd, inv): the view chains both ways (inv on unpadded rows)."""
        B, Hm, Wm, Ch = m.t.shape
        dh = Ch // heads
        ny, nx = Hm // wsz[0], Wm // wsz[1]

        def fwd(v):
            if shift:
                v = torch.roll(v, shifts=(-shift, -shift), dims=(1, 2))
            return v.reshape(B, ny, wsz[0], nx, wsz[1], heads, dh).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, dh)

        def inv(u):
            v = u.reshape(B, ny, nx, heads, wsz[0], wsz[1], dh).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, Hm, Wm, heads * dh)
            return torch.roll(v, shifts=(shift, shift), dims=(1, 2)) if shift else v
        return self.relayout(m, fwd, inv, pad_last=pad), fwd, inv

    def heads_windows(self, rows, fwd, inv, dh, pad):
        """the way back from window_heads' row form (rows [.., pad], the first dh columns used) to the NHWC map"""
        return self.relayout(rows, lambda u: inv(u[..., :dh]), lambda g: torch.nn.functional.pad(fwd(g), (0, pad - dh)))

    def add_rows_param(self, x, param, name):
        """x [B L, C] + param [1, L, C] for every sample (SwinIR's absolute position embedding, network_swinir.py:918-919);
        the parameter's gradient is the sum over the samples."""
        y = self.new(*x.t.shape)
        y.copy_(x.t)
        ops.add_periodic(y, param.data.reshape(-1).contiguous())
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is None:
                    return
                g = self._c(out.g)
                self.gparam(name, lambda o: ops.sum_periodic(g, o))
                self.acc(x, lambda o: o.copy_(g))
            self.back.append(bwd)
        return out

    def nearest_up2(self, x):
        """F.interpolate(scale_factor=2, mode='nearest') on an NHWC map; the adjoint adds the four copies."""
        xin = self._c(x.t)
        B, h, w, C = xin.shape
        y = self.new(B, 2 * h, 2 * w, C)
        ops.nearest_up2(xin, y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: ops.nearest_up2(o, self._c(out.g), adjoint=True))
            self.back.append(bwd)
        return out

    def conv_in1(self, x3, weight, bias, names):
        """first conv of a 1-channel image: x3 [B, H, W] -> [B, H, W, Co] (small.hip)."""
        B, H, W = x3.shape
        Co = weight.shape[0]
        y = self.new(B, H, W, Co)
        ops.conv3x3_cin1_fwd(x3, weight.data, None if bias is None else bias.data, Co, out=y)
        out = self._out(y)
        if self.save:
            def bwd(out=out):
                if out.g is None:
                    return
                dW, db = self._tmp(*weight.shape), self._tmp(Co)
                ops.conv3x3_cin1_wgrad(x3, out.g, dW, db)
                self.gparam(names[0], lambda o: o.copy_(dW))
                if names[1]:
                    self.gparam(names[1], lambda o: o.copy_(db))
            self.back.append(bwd)
        return out

    def conv_out1(self, x, weight, bias, names):
        """last conv to a 1-channel image: [B, H, W, Ci] -> Var of [B, H, W]."""
        B, H, W, Ci = x.t.shape
        y = self.new(B, H, W)
        ops.conv3x3_cout1_fwd(x.t if x.t.is_contiguous() else x.t.contiguous(), weight.data,
                              None if bias is None else bias.data, out=y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is None:
                    return
                dy = out.g
                xin = x.t if x.t.is_contiguous() else x.t.contiguous()
                dW = self._tmp(*weight.shape)
                ops.conv3x3_cin1_wgrad(dy, xin, dW, None, flip=True)
                self.gparam(names[0], lambda o: o.copy_(dW))
                if names[1]:
                    db = self._tmp(1)
                    ops.sum_into(dy, db)
                    self.gparam(names[1], lambda o: o.copy_(db))
                self.acc(x, lambda o: ops.conv3x3_cin1_fwd(dy, weight.data, None, Ci, out=o, flip=True))
            self.back.append(bwd)
        return out

    def prelu(self, x, alpha, name):
        y = self.new(*x.t.shape)
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        call("srhip_prelu_fwd", _p(xin), _p(alpha.data), _p(y), xin.numel(), _st())
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, xin=xin):
                if out.g is None:
                    return
                ws = ops.SCRATCH.get("prelu_ws", 4096, torch.float64, self.dev)
                da = self._tmp(1)
                dx = self._tmp(*x.t.shape)
                call("srhip_prelu_bwd", _p(out.g), _p(xin), _p(alpha.data), _p(dx), _p(da), _p(ws), xin.numel(), 0, _st())
                self.gparam(name, lambda o: o.copy_(da.view_as(o)))
                self.acc(x, lambda o: o.copy_(dx))
            self.back.append(bwd)
        return out

    def relu(self, x, slope=0.0):
        """ReLU (slope 0) or LeakyReLU(slope): out of place (the input may feed other ops)."""
        y = self.new(*x.t.shape)
        y.copy_(x.t)
        ops.leaky_relu_(y, slope)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out, y=y):
                if out.g is None:
                    return

                def prod(o):
                    o.copy_(out.g)
                    if slope == 0.0:
                        ops.relu_mask(o, y)
                    else:
                        ops.leaky_relu_mask(o, y, slope)
                self.acc(x, prod)
            self.back.append(bwd)
        return out

    def axpby(self, x, y, a=1.0, b=1.0):
        """a x + b y (torch.add / sub, residual connections)."""
        z = self.new(*x.t.shape)
        z.copy_(x.t)
        if a != 1.0:
            ops.axpby(z, z, a, 0.0)
        ops.axpby(z, y.t if y.t.is_contiguous() else y.t.contiguous(), b, 1.0)
        out = self._out(z)
        if self.save:
            def bwd(x=x, y=y, out=out):
                if out.g is None:
                    return
                self.acc(x, lambda o: (o.copy_(out.g), ops.axpby(o, o, a, 0.0) if a != 1.0 else None))
                self.acc(y, lambda o: (o.copy_(out.g), ops.axpby(o, o, b, 0.0) if b != 1.0 else None))
            self.back.append(bwd)
        return out

    def add_const(self, x, c):
        """x + c for a tensor c that takes no gradient (the interpolated input of SRFBN)."""
        z = self.new(*x.t.shape)
        z.copy_(x.t)
        ops.axpby(z, c, 1.0, 1.0)
        out = self._out(z)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: o.copy_(out.g))
            self.back.append(bwd)
        return out

    def cat(self, vs):
        """torch.cat(vs, channel dim): one buffer, the parts copied in by the strided axpby."""
        B, H, W = vs[0].t.shape[:3]
        Cs = [v.t.shape[3] for v in vs]
        z = self.new(B, H, W, sum(Cs))
        T = B * H * W
        o = 0
        for v, c in zip(vs, Cs):
            src = v.t
            call("srhip_axpby2d", _p(z) + 4 * o, z.stride(2), _p(src), src.stride(2), T, c, 1.0, 0.0, _st())
            o += c
        out = self._out(z)
        if self.save:
            def bwd(vs=vs, out=out):
                if out.g is None:
                    return
                o = 0
                for v, c in zip(vs, Cs):
                    def prod(dst, o=o, c=c):
                        call("srhip_axpby2d", _p(dst), dst.stride(2), _p(out.g) + 4 * o, out.g.stride(2), T, c, 1.0, 0.0, _st())
                    self.acc(v, prod)
                    o += c
            self.back.append(bwd)
        return out

    def shuffle(self, x, r):
        B, h, w, C = x.t.shape
        y = self.new(B, h * r, w * r, C // (r * r))
        ops.pixel_shuffle(x.t, r, nhwc_out=True, out=y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: ops.pixel_shuffle(out.g, r, nhwc_out=True, inverse=True, out=o))
            self.back.append(bwd)
        return out

    def unshuffle(self, x, r):
        B, Hh, Ww, C = x.t.shape
        y = self.new(B, Hh // r, Ww // r, C * r * r)
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        ops.pixel_shuffle(xin, r, nhwc_out=True, inverse=True, out=y)
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: ops.pixel_shuffle(out.g, r, nhwc_out=True, out=o))
            self.back.append(bwd)
        return out

    def pad_reflect(self, x):
        B, H, W, C = x.t.shape
        y = self.new(B, H + 2, W + 2, C)
        xin = x.t if x.t.is_contiguous() else x.t.contiguous()
        call("srhip_pad_reflect1", _p(xin), _p(y), B, H, W, C, 0, _st())
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: call("srhip_pad_reflect1", _p(out.g), _p(o), B, H, W, C, 1, _st()))
            self.back.append(bwd)
        return out

    def crop(self, x):
        B, Hp, Wp, C = x.t.shape
        H, W = Hp - 2, Wp - 2
        y = self.new(B, H, W, C)
        call("srhip_crop1", _p(x.t), _p(y), B, H, W, C, 0, _st())
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is not None:
                    self.acc(x, lambda o: call("srhip_crop1", _p(out.g), _p(o), B, H, W, C, 1, _st()))
            self.back.append(bwd)
        return out

    def crop1c(self, x):
        """crop 1 pixel per side of a 1-channel image [B, H+2, W+2] (the reflection-padded reconstruction conv): a view
        copy, its adjoint a zero-bordered placement -- plumbing on 1-channel images, no arithmetic."""
        B, Hp, Wp = x.t.shape
        y = self.new(B, Hp - 2, Wp - 2)
        y.copy_(x.t[:, 1:-1, 1:-1])
        out = self._out(y)
        if self.save:
            def bwd(x=x, out=out):
                if out.g is None:
                    return

                def prod(o):
                    o.zero_()
                    o[:, 1:-1, 1:-1].copy_(out.g)
                self.acc(x, prod)
            self.back.append(bwd)
        return out

    def backward(self, out, dy, grads):
        self.begin_backward(grads)
        out.g = dy
        for fn, vars_, bufs_ in reversed(self.back):
            fn()
            # the op's values have no reader left (their consumers' closures ran before this one): gradient and activation
            # buffers back to the pool
            for v in vars_:
                if v.gpool and v.g is not None:
                    self.pool.give(v.g)
                v.g, v.gpool = None, False
            for b in bufs_:
                self.pool.give(b)
        # a parameter the graph never reached keeps a zero gradient
        for k, gt in grads.items():
            if k not in self._written:
                gt.zero_()


class TapeEngine:
    """Base of the engines written as a tape graph: subclasses implement bank_entries() (which convs exist, in which
    form) and graph(tape, x3) -> output Var ([B, H, W] image)."""
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.bufs = _Bufs()
        self.bank = WeightBank()
        self.prepared = False
        self.saved = None

    def invalidate(self):
        self.prepared = False
        self._h16_ready = False

    def bucket_prefixes(self):
        return [[""]]           # one gradient bucket: every parameter

    def prepare(self):
        self.bank.begin()
        self.bank_entries()
        self.bank.finish(next(self.net.parameters()).device)
        self.prepared = True

    def forward(self, x, dp=None, save=True):
        """x [B, H, W] -> [B, 1, s H, s W]."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and hasattr(self, "forward_h16") and self._h16_ok():
            self.last_eval_path = "fp16 storage (attention blocks f32)"
            return self.forward_h16(x)
        tape = Tape(self.bufs, self.bank, save, x.device)
        out = self.graph(tape, x)
        if save:
            self.saved = (tape, out)
        B, H, W = out.t.shape
        return out.t.view(B, 1, H, W)

    def edsr_body_h16(self, x, layout, attention):
        """--amp evaluation of the EDSR-body nets (ENLCN, NLSN) on fp16 storage (conv_h16.hip): head conv, ResBlocks (two
        launches each: ReLU / residual as epilogues), the body's last conv with the long skip as its epilogue, the F -> 4F
        upsampler convs stored through their PixelShuffle(2), the output conv -- float16 feature maps, one fp16 product.
        The attention blocks (attention(tape, Var f32, body index) -> Var) run on f32 copies of their input, as under
        f32-storage --amp."""
        net, bank = self.net, self.bank
        F, rs = net.n_feats, float(net.res_scale)
        dev = x.device
        if not getattr(self, "_h16_ready", False):        # the upsampler convs whole, in sub-pixel-major column order
            tb = ops.PrepTable()
            self._h16_up = []
            for st in range(int(math.log2(net.upscale))):
                c = net.tail[0][2 * st]
                wp = ops.Bx3(9 * 4 * F, F, dev)
                tb.conv(c.weight.data, wp, ps2=True)
                self._h16_up.append((wp, c.bias.data))
            if self._h16_up:
                tb.build(dev).run()
            self._h16_ready = all(wp.fmt == 1 for wp, _ in self._h16_up)
            assert self._h16_ready, "edsr_body_h16: the upsampler conv does not take the fp16x2 operand format"
        t = Tape(self.bufs, bank, False, dev)
        h = ops.conv3x3_cin1_h16(x, net.head[0].weight.data, net.head[0].bias.data, F)
        res = h
        for i, kind in layout:
            if kind == "res":
                e0, e2 = bank.d[f"body.{i}.0"], bank.d[f"body.{i}.2"]
                a = ops.conv3x3_h16(res, e0.wp, e0.bias, F, epi=1)
                res = ops.conv3x3_h16(a, e2.wp, e2.bias, F, epi=2, R=res, alpha=rs)
            elif kind == "conv":
                e = bank.d[f"body.{i}"]
                res = ops.conv3x3_h16(res, e.wp, e.bias, F, epi=2, R=h)          # + the long skip
            else:
                res = attention(t, t.var(res.float(), need=False), i).t.half()
        for wp, b in self._h16_up:
            res = ops.conv3x3_h16(res, wp, b, 4 * F, ps2=True)
        y = ops.conv3x3_cout1_h16(res, net.tail[1].weight.data, net.tail[1].bias.data)
        B, H, W = y.shape
        return y.view(B, 1, H, W)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        assert self.saved is not None, "backward() without a saved forward"
        assert not need_dx, "tape engines: no gradient with respect to the input image"
        tape, out = self.saved
        tape.backward(out, dy.reshape(out.t.shape).contiguous(), grads)
        return None
