"""DRRN forward / backward as a fixed sequence of libsrhip launches (second of the plain CNNs).

Reference: dlib/models/network_drrn.py:22-126.  bicubic interpolation of the LR input (clamped); conv1 =
ReLU + conv 1->128; a RecursiveBlock that applies ONE residual unit (ReLU, conv, ReLU, conv; shared weights)
num_residual_units times, each time adding the block input; conv2 = ReLU + conv 128->1; + the interpolated
input.  No biases.  Two details of the reference carry over: its ReLUs are in place, so the first ReLU of the
first unit rectifies the block input itself -- the identity that every unit adds is relu(conv1 output), and
the ReLU in front of conv1 acts on an input that is already in [0, 1]:

    x0 = relu(conv1(xi));  r_0 = x0;  a_k = relu(conv_a(r_k));  r_{k+1} = relu(conv_b(a_k) + x0);
    y  = conv2(r_U) + xi

Launches: the 1-channel edge conv with a fused ReLU, 2U split-operand implicit-GEMM convs (epilogues: ReLU /
residual + ReLU, srhip_conv3x3_nhwc_split_ex epi 8), the 128->1 edge conv.  The weight gradients of the
shared convs are summed over the U applications.
"""
import torch
import torch.nn.functional as F

from . import ops
from .swinir_engine import _Bufs

CH = 128


class DRRNEngine:
    def __init__(self, net):
        self.net = net
        self.U = net.trunk.num_residual_unit
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
        """0.30 M parameters: one gradient bucket."""
        return [["conv1.", "trunk.", "conv2."]]

    def _convs(self):
        ru = self.net.trunk.residual_unit
        return [("wa", ru[1]), ("wb", ru[3])]

    def prepare(self):
        D, ws = self.derived, self.ws
        dev = self.net.conv2[1].weight.device
        if ws.use_bx3:
            sig = tuple(p.data_ptr() for p in self.net.parameters())
            if self._prep is None or sig != self._prep_sig:
                tb = ops.PrepTable()
                for name, conv in self._convs():
                    tb.conv(conv.weight.data, ws.planes(name + ".wp", 9 * CH, CH, dev))
                    tb.conv(conv.weight.data, ws.planes(name + ".wpt", 9 * CH, CH, dev), data_grad=True)
                self._prep, self._prep_sig = tb.build(dev), sig
            self._prep.run()
        else:
            for name, conv in self._convs():
                ops.pack_conv_weight(conv.weight.data, D.get(name + ".wp", 9, CH, CH, device=dev),
                                     D.get(name + ".wpt", 9, CH, CH, device=dev))
                ws.register(name + ".wp", D.d[name + ".wp"])
                ws.register(name + ".wpt", D.d[name + ".wpt"])
        self.prepared = True

    def interpolate(self, x):
        s = self.net.upscale
        out = F.interpolate(x, size=(s * x.shape[2], s * x.shape[3]), mode='bicubic', align_corners=False)
        return torch.clamp(out, min=0.0, max=1.0)[:, 0].contiguous()

    def forward_h16(self, x):
        """--amp evaluation on fp16 storage (conv_h16.hip): the same launches with float16 feature maps, one fp16 product."""
        net, U = self.net, self.U
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape

        def buf(name):
            return self.bufs.get("h." + name, B, H, W, CH, device=x.device, dtype=torch.float16)
        x0 = ops.conv3x3_cin1_h16(xi, net.conv1[1].weight.data, None, CH, out=buf("x0"), relu=True)
        r = x0
        for k in range(U):
            a = ops.conv3x3_h16(r, self.ws["wa.wp"], None, CH, out=buf("a"), epi=1)
            r = ops.conv3x3_h16(a, self.ws["wb.wp"], None, CH, out=buf(f"r{k % 2}"), epi=8, R=x0)
        y = ops.conv3x3_cout1_h16(r, net.conv2[1].weight.data, None, add=xi)
        return y.view(B, 1, H, W)

    def forward(self, x, dp=None, save=True):
        """x [B,H,W] (LR) -> [B,1,s*H,s*W]."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and self.ws.use_bx3 and self.ws["wa.wp"].fmt == 1:
            self.last_eval_path = "fp16 storage"
            return self.forward_h16(x)
        net, U = self.net, self.U
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape
        dev = x.device
        tag = "t" if save else "e"

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        x0 = buf("x0", B, H, W, CH)
        ops.conv3x3_cin1_fwd(xi, net.conv1[1].weight.data, None, CH, out=x0, relu=True)
        r = x0
        rs, as_ = [], []
        for k in range(U):
            a = buf(f"a{k if save else 0}", B, H, W, CH)
            ops.conv3x3(r, self.ws["wa.wp"], None, CH, out=a, epi=1)
            rn = buf(f"r{k + 1 if save else 1 + k % 2}", B, H, W, CH)
            if self.ws.use_bx3:                              # r_{k+1} = relu(conv_b(a_k) + x0): residual + ReLU as the epilogue
                ops.conv3x3(a, self.ws["wb.wp"], None, CH, out=rn, epi=8, R=x0)
            else:
                ops.conv3x3(a, self.ws["wb.wp"], None, CH, out=rn, epi=2, R=x0)
                ops.relu_mask(rn, rn)
            if save:
                rs.append(r)
                as_.append(a)
            r = rn
        y = torch.empty(B, H, W, device=dev) if not save else buf("y", B, H, W)
        ops.conv3x3_cout1_fwd(r, net.conv2[1].weight.data, None, out=y)
        ops.axpby(y, xi, 1.0, 1.0)
        if save:
            self.saved = dict(xi=xi, x0=x0, rs=rs, as_=as_, r_last=r, B=B, H=H, W=W)
        return y.view(B, 1, H, W)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        assert not need_dx, "DRRN (libsrhip): no gradient through the bicubic interpolation of the input"
        net, U = self.net, self.U
        B, H, W = sv["B"], sv["H"], sv["W"]
        dev = dy.device

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        dy = dy.reshape(B, H, W).contiguous()
        ops.conv3x3_cin1_wgrad(dy, sv["r_last"], grads["conv2.1.weight"], None, flip=True)
        gu, ga, gx0 = buf("gu", B, H, W, CH), buf("ga", B, H, W, CH), buf("gx0", B, H, W, CH)
        ops.conv3x3_cin1_fwd(dy, net.conv2[1].weight.data, None, CH, out=gu, flip=True)
        ops.relu_mask(gu, sv["r_last"])                      # through r_U = relu(u_U)
        dWa, dWb = grads["trunk.residual_unit.1.weight"], grads["trunk.residual_unit.3.weight"]
        tWa, tWb = buf("tWa", CH, CH, 3, 3), buf("tWb", CH, CH, 3, 3)
        gx0.zero_()
        for k in reversed(range(U)):                         # gu = d / d u_{k+1}, u_{k+1} = conv_b(a_k) + x0
            ops.axpby(gx0, gu, 1.0, 1.0)                     # the identity path of every unit
            first = k == U - 1
            ops.conv3x3_wgrad(gu, sv["as_"][k], dWb if first else tWb, None)
            ops.conv3x3(gu, self.ws["wb.wpt"], None, CH, out=ga, epi=4, R=sv["as_"][k])     # * (a_k > 0)
            ops.conv3x3_wgrad(ga, sv["rs"][k], dWa if first else tWa, None)
            if not first:                                    # shared weights: sum over the applications
                ops.axpby(dWb, tWb, 1.0, 1.0)
                ops.axpby(dWa, tWa, 1.0, 1.0)
            # d / d r_k, masked by r_k > 0: for k >= 1 that is the unit's ReLU (-> d / d u_k); for k = 0
            # it is the in-place ReLU that made x0 out of conv1's output
            ops.conv3x3(ga, self.ws["wa.wpt"], None, CH, out=gu, epi=4, R=sv["rs"][k])
        ops.relu_mask(gx0, sv["x0"])
        ops.axpby(gu, gx0, 1.0, 1.0)                         # d / d conv1 output
        ops.conv3x3_cin1_wgrad(sv["xi"], gu, grads["conv1.1.weight"], None)
        # single bucket = the last one: TrainStep's reducer sends it after backward (announcing it
        # here too reduced it twice -- the sum instead of the mean -- before the reducer tracked
        # which buckets were done)
        return None
