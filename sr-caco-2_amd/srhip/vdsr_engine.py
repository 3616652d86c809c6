"""VDSR forward / backward as a fixed sequence of libsrhip launches (the plain CNNs reuse the
3x3-conv kernels).

Reference: dlib/models/network_vdsr.py:24-126 -- bicubic interpolation of the LR input to the HR size
(clamped to [0, 1]); conv1 1->64 + ReLU; 18 x (conv 64->64 + ReLU); conv2 64->1; + the interpolated input.
No biases.  The interpolation stays on stock PyTorch-ROCm (F.interpolate, as the reference); every conv is a
libsrhip kernel: the 1-channel edge convs of small.hip, the bf16x3 implicit-GEMM conv with the ReLU as its
epilogue, the ReLU mask of the backward as the data-gradient conv's epilogue (epi 4).  NHWC throughout.
"""
import torch
import torch.nn.functional as F

from . import ops
from .swinir_engine import _Bufs

CH = 64


class VDSREngine:
    def __init__(self, net):
        self.net = net
        self.nt = len(net.trunk)
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()
        self.ws.use_bx3 = ops.bx3_nt_for(CH)
        self._prep = self._prep_sig = None
        self.prepared = False
        self.saved = None

    def invalidate(self):
        self.prepared = False

    def bucket_prefixes(self):
        """0.67 M parameters: one gradient bucket."""
        return [["conv1.", "trunk.", "conv2."]]

    def prepare(self):
        D, ws = self.derived, self.ws
        dev = self.net.conv2.weight.device
        convs = [(f"t{k}", self.net.trunk[k].conv) for k in range(self.nt)]
        if ws.use_bx3:
            sig = tuple(p.data_ptr() for p in self.net.parameters())
            if self._prep is None or sig != self._prep_sig:
                tb = ops.PrepTable()
                for name, conv in convs:
                    tb.conv(conv.weight.data, ws.planes(name + ".wp", 9 * CH, CH, dev))
                    tb.conv(conv.weight.data, ws.planes(name + ".wpt", 9 * CH, CH, dev), data_grad=True)
                self._prep, self._prep_sig = tb.build(dev), sig
            self._prep.run()
        else:
            for name, conv in convs:
                ops.pack_conv_weight(conv.weight.data, D.get(name + ".wp", 9, CH, CH, device=dev),
                                     D.get(name + ".wpt", 9, CH, CH, device=dev))
                ws.register(name + ".wp", D.d[name + ".wp"])
                ws.register(name + ".wpt", D.d[name + ".wpt"])
        self.prepared = True

    def interpolate(self, x):
        """network_vdsr.py:78-90: [B,1,h,w] -> [B,s*h,s*w], bicubic (align_corners False), clamped."""
        s = self.net.upscale
        out = F.interpolate(x, size=(s * x.shape[2], s * x.shape[3]), mode='bicubic', align_corners=False)
        return torch.clamp(out, min=0.0, max=1.0)[:, 0].contiguous()

    def forward_h16(self, x):
        """--amp evaluation on fp16 storage (conv_h16.hip): the same launches with float16 feature maps, one fp16 product."""
        net = self.net
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape
        a = self.bufs.get("h.a0", B, H, W, CH, device=x.device, dtype=torch.float16)
        ops.conv3x3_cin1_h16(xi, net.conv1[0].weight.data, None, CH, out=a, relu=True)
        for k in range(self.nt):
            an = self.bufs.get(f"h.a{1 + k % 2}", B, H, W, CH, device=x.device, dtype=torch.float16)
            ops.conv3x3_h16(a, self.ws[f"t{k}.wp"], None, CH, out=an, epi=1)
            a = an
        y = ops.conv3x3_cout1_h16(a, net.conv2.weight.data, None, add=xi)       # + the interpolated input
        return y.view(B, 1, H, W)

    def forward(self, x, dp=None, save=True):
        """x [B,H,W] (LR) -> [B,1,s*H,s*W]."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and self.ws.use_bx3 and self.ws["t0.wp"].fmt == 1:
            self.last_eval_path = "fp16 storage"
            return self.forward_h16(x)
        net = self.net
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape
        dev = x.device
        tag = "t" if save else "e"

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        a = buf("a0", B, H, W, CH)
        ops.conv3x3_cin1_fwd(xi, net.conv1[0].weight.data, None, CH, out=a, relu=True)
        acts = [a]
        for k in range(self.nt):
            an = buf(f"a{k + 1 if save else 1 + k % 2}", B, H, W, CH)
            ops.conv3x3(a, self.ws[f"t{k}.wp"], None, CH, out=an, epi=1)
            if save:
                acts.append(an)
            a = an
        y = torch.empty(B, H, W, device=dev) if not save else buf("y", B, H, W)
        ops.conv3x3_cout1_fwd(a, net.conv2.weight.data, None, out=y)
        ops.axpby(y, xi, 1.0, 1.0)                       # + the interpolated input (global residual)
        if save:
            self.saved = dict(xi=xi, acts=acts, B=B, H=H, W=W)
        return y.view(B, 1, H, W)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        assert not need_dx, "VDSR (libsrhip): no gradient through the bicubic interpolation of the input"
        net = self.net
        B, H, W = sv["B"], sv["H"], sv["W"]
        dev = dy.device
        acts = sv["acts"]

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        dy = dy.reshape(B, H, W).contiguous()
        # conv2 (64 -> 1): weight gradient = the 1-channel kernel with x / dy swapped and flipped taps,
        # data gradient = the 1-channel forward kernel with flipped taps; then the last ReLU's mask
        ops.conv3x3_cin1_wgrad(dy, acts[-1], grads["conv2.weight"], None, flip=True)
        ga, gb = buf("ga", B, H, W, CH), buf("gb", B, H, W, CH)
        g = ga
        ops.conv3x3_cin1_fwd(dy, net.conv2.weight.data, None, CH, out=g, flip=True)
        ops.relu_mask(g, acts[-1])
        for k in reversed(range(self.nt)):               # a_{k+1} = relu(conv_k(a_k)); g = d/d(conv_k output)
            ops.conv3x3_wgrad(g, acts[k], grads[f"trunk.{k}.conv.weight"], None)
            other = gb if g is ga else ga
            ops.conv3x3(g, self.ws[f"t{k}.wpt"], None, CH, out=other, epi=4, R=acts[k])     # * (a_k > 0)
            g = other
        ops.conv3x3_cin1_wgrad(sv["xi"], g, grads["conv1.0.weight"], None)
        # single bucket = the last one: TrainStep's reducer sends it after backward (announcing it
        # here too reduced it twice -- the sum instead of the mean -- before the reducer tracked
        # which buckets were done)
        return None
