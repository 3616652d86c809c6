"""SRCNN forward / backward as libsrhip launches.

Reference: dlib/models/network_srcnn.py:23-69 (the registered class, select_network.py:207-210):
features = Conv2d(1, 1024, 5, 1, 2) + ReLU, map = Conv2d(1024, 128, 1) + ReLU, reconstruction =
Conv2d(128, 1, 1), applied to the low-resolution image ALREADY interpolated to the target size (the
batch's 'l_to_h_img', model_plain.py:184-195).  Every layer is a token-matrix GEMM here: the 5x5
1-channel conv through its patch matrix [T x 28] (srhip_im2col_c1, 25 taps + 3 zero columns), the 1x1
convs directly on [T x C]; ReLU is the GEMM epilogue, its mask the epilogue of the data-gradient GEMM
(epi 4), weight gradients run on the grouped TN kernel.  The 1-wide output layer is padded to 4 columns
(zero rows of W3), so that every matrix meets the 4-float alignment of the kernels."""
import torch

from . import ops
from .swinir_engine import _Bufs

C1, C2, KP1 = 1024, 128, 28      # channels; padded taps of the 5x5 layer


class SRCNNEngine:
    def __init__(self, net):
        self.net = net
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()
        self.ws.use_bx3 = True
        self._prep = self._prep_sig = None
        self.prepared = False
        self.saved = None
        self.rows_max = ((1 << 29) - 1) // C1      # rows of a 1024-wide operand one GEMM call addresses (32-bit element offsets)

    def invalidate(self):
        self.prepared = False
        self._h16_ready = False

    def bucket_prefixes(self):
        return [["features.", "map.", "reconstruction."]]

    def prepare(self):
        net, D, ws = self.net, self.derived, self.ws
        dev = net.map[0].weight.device
        w1p = D.get("w1p", C1, KP1, device=dev)          # [1024][25] taps + 3 zero columns
        w3p = D.get("w3p", 4, C2, device=dev)            # [1][128] + 3 zero rows
        sig = tuple(p.data_ptr() for p in net.parameters())
        if self._prep is None or sig != self._prep_sig:
            w1p.zero_()
            w3p.zero_()
            tb = ops.PrepTable()
            tb.linear(w1p, ws.planes("w1", C1, KP1, dev))
            tb.linear(net.map[0].weight.data.view(C2, C1), ws.planes("w2", C2, C1, dev))
            tb.linear(net.map[0].weight.data.view(C2, C1), ws.planes("w2T", C1, C2, dev), transpose=True)
            tb.linear(w3p, ws.planes("w3", 4, C2, dev))
            tb.linear(w3p, ws.planes("w3T", C2, 4, dev), transpose=True)
            self._prep, self._prep_sig = tb.build(dev), sig
        w1p[:, :25].copy_(net.features[0].weight.data.view(C1, 25))
        w3p[:1].copy_(net.reconstruction.weight.data.view(1, C2))
        self._prep.run()
        self.prepared = True

    def _prepare_h16(self, dev):
        """the three layers as centre-tap 3x3 operands of the fp16-storage conv (conv_h16.hip): 32 (25 taps + 7 zero) ->
        1024, 1024 -> 128, and the 128 -> 1 tail conv"""
        net, D = self.net, self.derived
        w1 = D.get("h.w1", C1, 32, 3, 3, device=dev)
        w1.zero_()
        w1[:, :25, 1, 1].copy_(net.features[0].weight.data.view(C1, 25))
        w2 = D.get("h.w2", C2, C1, 3, 3, device=dev)
        w2.zero_()
        w2[:, :, 1, 1].copy_(net.map[0].weight.data.view(C2, C1))
        w3 = D.get("h.w3", 1, C2, 3, 3, device=dev)
        w3.zero_()
        w3[:, :, 1, 1].copy_(net.reconstruction.weight.data.view(1, C2))
        tb = ops.PrepTable()
        self._h16_w1, self._h16_w2 = ops.Bx3(9 * C1, 32, dev), ops.Bx3(9 * C2, C1, dev)
        tb.conv(w1, self._h16_w1, force_f16=True)
        tb.conv(w2, self._h16_w2, force_f16=True)
        tb.build(dev).run()
        self._h16_ready = self._h16_w1.fmt == 1 and self._h16_w2.fmt == 1
        return self._h16_ready

    def forward_h16(self, x):
        """--amp evaluation: the three layers in ONE kernel (srhip_srcnn_fwd_h16, conv_h16.hip) -- the 1024-channel feature map
        (8.6 GB at B = 8, 512 x 512 in f32: the layer-wise forward is its write and read-back) is walked in 64-channel chunks
        through LDS and never reaches HBM; fp16 products, f32 accumulate."""
        net, D = self.net, self.derived
        B, H, W = x.shape
        dev = x.device
        T = B * H * W
        y = torch.empty(B, H, W, device=dev)
        ops.srcnn_fwd_h16(None, self._h16_w1, net.features[0].bias.data, self._h16_w2, net.map[0].bias.data,
                          net.reconstruction.weight.data.view(C2), net.reconstruction.bias.data, y.view(T),
                          image=x if x.is_contiguous() else x.contiguous())
        return y.view(B, 1, H, W)

    def forward(self, x, dp=None, save=True):
        """x [B,H,W] (already at the target size) -> [B,1,H,W]."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and (getattr(self, "_h16_ready", False) or self._prepare_h16(x.device)):
            self.last_eval_path = "fp16 storage"
            return self.forward_h16(x)
        net, D, ws = self.net, self.derived, self.ws
        B, H, W = x.shape
        T = B * H * W
        dev = x.device
        tag = "t" if save else "e"

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        b3 = D.get("b3p", 4, device=dev)
        b3.zero_()
        b3[:1].copy_(net.reconstruction.bias.data)
        y = torch.empty(B, H, W, device=dev) if not save else buf("y", B, H, W)
        # the GEMM kernels address an operand with 32-bit element offsets (rows * 1024 < 2^29): the batch is walked in groups
        # of images that fit (self.rows_max rows per call); a training step keeps every group's activations in whole-batch
        # buffers and its backward walks the same groups, adding their weight gradients
        per = max(1, self.rows_max // (H * W))
        if save:
            a0w, h1w, h2w = buf("a0", T, KP1), buf("h1", T, C1), buf("h2", T, C2)
        for b0 in range(0, B, per):
            nb = min(per, B - b0)
            t = nb * H * W
            r0 = b0 * H * W
            a0 = ops.im2col_c1(x[b0:b0 + nb], 5, KP1, out=a0w[r0:r0 + t] if save else buf("a0", t, KP1))
            h1 = ops.gemm_nt(a0, ws["w1"], net.features[0].bias.data, out=h1w[r0:r0 + t] if save else buf("h1", t, C1), epi=1)
            h2 = ops.gemm_nt(h1, ws["w2"], net.map[0].bias.data, out=h2w[r0:r0 + t] if save else buf("h2", t, C2), epi=1)
            y4 = ops.gemm_nt(h2, ws["w3"], b3, out=buf("y4", min(per, B) * H * W, 4)[:t])
            y[b0:b0 + nb].view(t).copy_(y4[:, 0])
        if save:
            self.saved = dict(a0=a0w, h1=h1w, h2=h2w, B=B, H=H, W=W, per=per)
        return y.view(B, 1, H, W)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        assert not need_dx, "SRCNN (libsrhip): no gradient with respect to the interpolated input"
        net, D, ws = self.net, self.derived, self.ws
        B, H, W = sv["B"], sv["H"], sv["W"]
        T = B * H * W
        dev = dy.device

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        dy4 = buf("dy4", T, 4)
        dy4.zero_()
        dy4[:, 0].copy_(dy.reshape(T))
        dw3, db3 = buf("dw3", 4, C2), buf("db3", 4)
        dw1 = buf("dw1", C1, KP1)
        step = sv["per"] * H * W
        for r0 in range(0, T, step):
            r1 = min(T, r0 + step)
            first = r0 == 0
            # (scratch at the FULL group size, sliced for a smaller last group: a changing shape would reallocate 2 GiB twice
            # per step -- no allocation in the steady state, capturable)
            dh2 = ops.gemm_nt(dy4[r0:r1], ws["w3T"], None, out=buf("dh2", min(step, T), C2)[:r1 - r0], epi=4, R=sv["h2"][r0:r1])   # * (h2 > 0)
            dh1 = ops.gemm_nt(dh2, ws["w2T"], None, out=buf("dh1", min(step, T), C1)[:r1 - r0], epi=4, R=sv["h1"][r0:r1])          # * (h1 > 0)
            # the first group writes the gradients, the others add theirs
            tw3, tb3, tw1 = (dw3, db3, dw1) if first else (buf("dw3.t", 4, C2), buf("db3.t", 4), buf("dw1.t", C1, KP1))
            tw2 = grads["map.0.weight"].view(C2, C1) if first else buf("dw2.t", C2, C1)
            tb2 = grads["map.0.bias"] if first else buf("db2.t", C2)
            tb1 = grads["features.0.bias"] if first else buf("db1.t", C1)
            ops.linear_wgrad_grouped([
                dict(dY=dy4[r0:r1], X=sv["h2"][r0:r1], dW=tw3, db=tb3),
                dict(dY=dh2, X=sv["h1"][r0:r1], dW=tw2, db=tb2),
                dict(dY=dh1, X=sv["a0"][r0:r1], dW=tw1, db=tb1),
            ])
            if not first:
                for dst, src in ((dw3, tw3), (db3, tb3), (dw1, tw1), (grads["map.0.weight"].view(C2, C1), tw2),
                                 (grads["map.0.bias"], tb2), (grads["features.0.bias"], tb1)):
                    ops.axpby(dst, src, 1.0, 1.0)
        grads["reconstruction.weight"].view(1, C2).copy_(dw3[:1])
        grads["reconstruction.bias"].copy_(db3[:1])
        grads["features.0.weight"].view(C1, 25).copy_(dw1[:, :25])
        return None
