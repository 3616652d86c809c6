"""EDSR-baseline forward / backward as a fixed sequence of libsrhip launches.

Wiring of the reference's EDSR blocks (dlib/models/network_nlsn.py:38-128 blocks,
:355-369 forward without the attention modules; sizes from
dlib/utils/utils_init_default_args.py:37-50): head conv 1->F, N x ResBlock
(conv-ReLU-conv, *res_scale, +x), conv + long skip, Upsampler (conv F->4F +
PixelShuffle(2) per octave), tail conv F->1.  NHWC throughout; ReLU, res_scale
and both residual adds are conv epilogues, the ReLU mask is the epilogue of the
data-gradient conv, pixel shuffles are index kernels.
"""
import math
import os

import torch

from . import ops
from .swinir_engine import _Bufs


class EDSREngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.F = net.n_feats
        self.nb = net.n_resblocks
        self.stages = int(math.log2(net.scale))
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()
        # conv forward / data gradient on the bf16x3 kernels from 64 features on; the weight gradients of a
        # 64-feature net stay on the exact-f32 MFMA kernels (ops.conv3x3_wgrad decides by ops.bx3_for)
        self.ws.use_bx3 = ops.bx3_nt_for(self.F)
        self._prep = self._prep_sig = None
        self.prepared = False
        # weight gradients on a side stream beside the data-gradient chain (backward): SRHIP_EDSR_SIDE_WGRAD=1.  Measured
        # (round 5, EDSR x8, B = 8, same box, two rounds): 1715.2 / 1715.0 patches/s on one stream, 1687.3 / 1671.7 with the
        # side stream -- the chain's one-block-per-CU launches do not leave the weight-gradient blocks room they could use
        # (both slow down); off.
        self.side_wgrad = os.environ.get("SRHIP_EDSR_SIDE_WGRAD", "0") == "1"
        self.wstream = None
        self.saved = None
        # Upsampler stage = conv F -> 4F + PixelShuffle(2) as one kernel per direction (no [B,H,W,4F] tensor,
        # no shuffle launches); SRHIP_FUSE_PS=0: conv + index kernel
        self.fuse_ps = self.ws.use_bx3 and ops.ps2_fusable(self.F, 4 * self.F) and \
            os.environ.get("SRHIP_FUSE_PS", "1") != "0"
        # One launch per ResBlock and direction (resblock.hip: conv -> ReLU -> conv -> + skip from LDS, round 6) instead of two:
        # SRHIP_FUSE_RESBLOCK=0 keeps the two conv launches.  64-feature nets on the fp16x2 conv operands only.
        self.fuse_rb = self.ws.use_bx3 and ops.resblock64_fusable(self.F) and \
            os.environ.get("SRHIP_FUSE_RESBLOCK", "1") != "0"
        # ... where a conv launch is latency-bound: the fused kernel recomputes the first conv on a one-pixel ring around its
        # 4 x 16 tile (108 mid pixels for 64), which a launch that fills the chip many times over pays in matrix-core time
        self.fuse_rb_maxpix = int(os.environ.get("SRHIP_FUSE_RESBLOCK_MAXPIX", str(8 * 64 * 64)))

    def invalidate(self):
        self.prepared = False

    def bucket_prefixes(self):
        """EDSR-baseline has 1.4-1.7 M parameters (5.5-6.7 MB): one gradient bucket."""
        return [["head.", "body.", "tail."]]

    def _body_convs(self):
        net = self.net
        for k in range(self.nb):
            yield f"b{k}.0", net.body[k].body[0]
            yield f"b{k}.2", net.body[k].body[2]
        yield "bend", net.body[self.nb]
        for i in range(self.stages):
            yield f"up{i}", net.tail[0][2 * i]

    def prepare(self):
        D, ws = self.derived, self.ws
        dev = self.net.head[0].weight.device
        if ws.use_bx3:      # all conv packs as bf16x3 planes by one launch
            sig = tuple(p.data_ptr() for p in self.net.parameters())
            if self._prep is None or sig != self._prep_sig:
                tb = ops.PrepTable()
                for name, conv in self._body_convs():
                    co, ci = conv.weight.shape[:2]
                    ps2 = self.fuse_ps and name.startswith("up")
                    tb.conv(conv.weight.data, ws.planes(name + ".wp", 9 * co, ci, dev), ps2=ps2)
                    tb.conv(conv.weight.data, ws.planes(name + ".wpt", 9 * ci, co, dev), data_grad=True, ps2=ps2)
                self._prep, self._prep_sig = tb.build(dev), sig
            self._prep.run()
        else:
            for name, conv in self._body_convs():
                co, ci = conv.weight.shape[:2]
                ops.pack_conv_weight(conv.weight.data, D.get(name + ".wp", 9, co, ci, device=dev),
                                     D.get(name + ".wpt", 9, ci, co, device=dev))
                ws.register(name + ".wp", D.d[name + ".wp"])
                ws.register(name + ".wpt", D.d[name + ".wpt"])
        self.prepared = True

    # ------------------------------------------------------------------ forward
    def _h16_ok(self):
        return (self.ws.use_bx3 and self.fuse_ps and self.F % 64 == 0
                and all(self.ws[n + ".wp"].fmt == 1 for n, _ in self._body_convs()))

    def forward_h16(self, x):
        """--amp evaluation on fp16 storage (conv_h16.hip): the same launches with float16 feature maps, one fp16 product;
        the upsampler's conv stores through its PixelShuffle(2)."""
        net, F = self.net, self.F
        B, H, W = x.shape
        dev = x.device

        def buf(name, *shape):
            return self.bufs.get("h." + name, *shape, device=dev, dtype=torch.float16)
        f0 = ops.conv3x3_cin1_h16(x, net.head[0].weight.data, net.head[0].bias.data, F, out=buf("f0", B, H, W, F))
        r, rs = f0, float(net.res_scale)
        for k in range(self.nb):
            a = ops.conv3x3_h16(r, self.ws[f"b{k}.0.wp"], net.body[k].body[0].bias.data, F, out=buf("a", B, H, W, F), epi=1)
            r = ops.conv3x3_h16(a, self.ws[f"b{k}.2.wp"], net.body[k].body[2].bias.data, F, out=buf(f"r{k % 2}", B, H, W, F),
                                epi=2, R=r, alpha=rs)
        u = ops.conv3x3_h16(r, self.ws["bend.wp"], net.body[self.nb].bias.data, F, out=buf("rb", B, H, W, F), epi=2, R=f0)
        h, w = H, W
        for i in range(self.stages):
            u = ops.conv3x3_h16(u, self.ws[f"up{i}.wp"], net.tail[0][2 * i].bias.data, 4 * F, out=buf(f"u{i}", B, 2 * h, 2 * w, F),
                                ps2=True)
            h, w = 2 * h, 2 * w
        y = ops.conv3x3_cout1_h16(u, net.tail[1].weight.data, net.tail[1].bias.data)
        return y.view(B, 1, h, w)

    def forward(self, x, dp=None, save=True):
        """x [B,H,W] -> [B,1,s*H,s*W]."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and self._h16_ok():
            self.last_eval_path = "fp16 storage"
            return self.forward_h16(x)
        net, F, D = self.net, self.F, self.derived
        B, H, W = x.shape
        dev = x.device
        tag = "t" if save else "e"

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        f0 = buf("f0", B, H, W, F)
        ops.conv3x3_cin1_fwd(x, net.head[0].weight.data, net.head[0].bias.data, F, out=f0)
        r = f0
        blocks = []
        rs = float(net.res_scale)
        fuse_rb = self.fuse_rb and not ops.lib.srhip_get_matmul_mode() and B * H * W <= self.fuse_rb_maxpix and \
            ops.resblock64_fusable(F, self.ws["b0.0.wp"] if self.nb else None)
        for k in range(self.nb):
            kk = k if save else k % 2
            a = buf(f"a{kk if save else 0}", B, H, W, F)
            rn = buf(f"r{kk}", B, H, W, F)
            if fuse_rb:
                ops.resblock64_fwd(r, self.ws[f"b{k}.0.wp"], net.body[k].body[0].bias.data, self.ws[f"b{k}.2.wp"],
                                   net.body[k].body[2].bias.data, rs, a, rn)
            else:
                ops.conv3x3(r, self.ws[f"b{k}.0.wp"], net.body[k].body[0].bias.data, F, out=a, epi=1)
                ops.conv3x3(a, self.ws[f"b{k}.2.wp"], net.body[k].body[2].bias.data, F, out=rn, epi=2, R=r,
                            alpha=rs)
            if save:
                blocks.append((r, a))
            r = rn
        rb = buf("rb", B, H, W, F)
        ops.conv3x3(r, self.ws["bend.wp"], net.body[self.nb].bias.data, F, out=rb, epi=2, R=f0)
        u, h, w = rb, H, W
        ups = []
        for i in range(self.stages):
            un = buf(f"u{i}", B, 2 * h, 2 * w, F)
            if self.fuse_ps:
                ops.conv3x3_ps2(u, self.ws[f"up{i}.wp"], net.tail[0][2 * i].bias.data, un)
            else:
                c = buf(f"c{i}", B, h, w, 4 * F)
                ops.conv3x3(u, self.ws[f"up{i}.wp"], net.tail[0][2 * i].bias.data, 4 * F, out=c)
                ops.pixel_shuffle(c, 2, nhwc_out=True, out=un)
            if save:
                ups.append(u)
            u, h, w = un, 2 * h, 2 * w
        y = torch.empty(B, h, w, device=dev) if not save else buf("y", B, h, w)
        ops.conv3x3_cout1_fwd(u, net.tail[1].weight.data, net.tail[1].bias.data, out=y)
        if save:
            self.saved = dict(x=x, f0=f0, blocks=blocks, r_last=r, ups=ups, u_last=u, B=B, H=H, W=W)
        return y.view(B, 1, h, w)

    # ------------------------------------------------------------------ backward
    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        net, F, D = self.net, self.F, self.derived
        B, H, W = sv["B"], sv["H"], sv["W"]
        dev = dy.device
        rs = float(net.res_scale)

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        def G(name):
            return grads[name]

        # Round 5 experiment (off by default, see __init__): the WEIGHT gradients on a side stream beside the data-gradient chain.  The chain is what the next
        # kernel waits for -- 2 nb + 2 launches of one block per CU, each ~16 us of latency for 3 us of matrix work at the
        # x8 patch size (64 x 64 LR pixels) -- and no weight gradient is on it: each needs only its layer's incoming gradient
        # and saved input, both complete when the chain has passed the layer.  side(fn): everything enqueued so far is fn's
        # input (one event), fn's launches go to the side stream; the streams join at the end (and inside a captured step:
        # a fork / join of the graph).  All weight-gradient launches share that one stream, so their scratch buffers
        # (ops.SCRATCH: partial sums) are used strictly one launch after the other, as before.
        main = torch.cuda.current_stream()
        batched = ops.bx3_for(F, F) and self.nb > 0          # (the deferred form: every layer's operands in buffers of their own)
        use_side = self.side_wgrad and dev.type == "cuda" and batched
        if use_side and self.wstream is None:
            self.wstream = torch.cuda.Stream(device=dev)

        def side(fn):
            if not use_side:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            self.wstream.wait_event(ev)
            with torch.cuda.stream(self.wstream):
                fn()

        s = net.scale
        dy = dy.reshape(B, H * s, W * s).contiguous()
        # tail conv F->1: weight grad = 1-channel wgrad kernel with the roles of
        # x / dy swapped and flipped taps; data grad = 1-channel fwd kernel, flipped
        side(lambda: ops.conv3x3_cin1_wgrad(dy, sv["u_last"], G("tail.1.weight"), None, flip=True))
        ops.sum_into(dy, G("tail.1.bias"))
        h, w = H * s, W * s
        du = buf(f"du{self.stages}", B, h, w, F)
        ops.conv3x3_cin1_fwd(dy, net.tail[1].weight.data, None, F, out=du, flip=True)
        for i in reversed(range(self.stages)):
            h, w = h // 2, w // 2
            if self.fuse_ps:     # both gradients read the gradient of the shuffled image
                side(lambda du=du, i=i: ops.conv3x3_wgrad(du, sv["ups"][i], G(f"tail.0.{2 * i}.weight"),
                                                          G(f"tail.0.{2 * i}.bias"), ps2=True))
                dun = buf(f"du{i}", B, h, w, F)
                ops.conv3x3_ps2_bwd_data(du, self.ws[f"up{i}.wpt"], dun)
                du = dun
            else:
                dc = buf(f"dc{i}", B, h, w, 4 * F)
                ops.pixel_shuffle(du, 2, nhwc_out=True, inverse=True, out=dc)
                side(lambda dc=dc, i=i: ops.conv3x3_wgrad(dc, sv["ups"][i], G(f"tail.0.{2 * i}.weight"),
                                                          G(f"tail.0.{2 * i}.bias")))
                du = buf(f"du{i}", B, h, w, F)
                ops.conv3x3(dc, self.ws[f"up{i}.wpt"], None, F, out=du)
        drb = du                                             # grad wrt rb (= also grad wrt f0 via the skip)
        # The body's weight gradients are DEFERRED: every layer keeps its incoming gradient in a buffer
        # of its own (2*nb+1 buffers of B*H*W*F floats; sized for 288 GB of HBM) and all 2*nb+1
        # problems -- one shape -- go through ONE batched contraction + ONE reducer launch after the
        # data-gradient chain.  Alone, a 64 -> 64 problem at 64x64x8 pixels is 2.4 GFLOP: it was cut in
        # ~85 reduce slices to fill the chip and paid a 12.5 MB partial buffer + a reducer per layer
        # (rocprofv3, EDSR x8: 36 x (96 + 24) us of a 9.6 ms step).
        wg = []                                              # (dY, X, dW, db)

        # (with the side stream the batch leaves in groups of >= `grp` problems as the chain passes them: each group runs
        # beside the rest of the chain; without it: one launch at the end, as in round 2)
        grp = max(8, (2 * self.nb + 1 + 3) // 4) if use_side else 1 << 30

        def flush():
            if wg:
                items = list(wg)
                wg.clear()
                side(lambda: ops.conv3x3_wgrad_batched(items))

        def wgrad(dY, X, wname, bname):
            if batched:
                wg.append((dY, X, G(wname), G(bname)))
                if len(wg) >= grp:
                    flush()
            else:
                side(lambda: ops.conv3x3_wgrad(dY, X, G(wname), G(bname)))

        wgrad(drb, sv["r_last"], f"body.{self.nb}.weight", f"body.{self.nb}.bias")
        if batched:
            gs = [buf(f"g{k}", B, H, W, F) for k in range(self.nb + 1)]      # gs[k]: gradient wrt block k's input
            das = [buf(f"da{k}", B, H, W, F) for k in range(self.nb)]
        else:
            ga, gb, da1 = buf("ga", B, H, W, F), buf("gb", B, H, W, F), buf("da", B, H, W, F)
            gs = [ga if (self.nb - k) % 2 == 0 else gb for k in range(self.nb + 1)]
            das = [da1] * self.nb
        g = gs[self.nb]
        fuse_rb = self.fuse_rb and not ops.lib.srhip_get_matmul_mode() and B * H * W <= self.fuse_rb_maxpix and \
            ops.resblock64_fusable(F, self.ws["b0.0.wpt"] if self.nb else None) and batched
        ops.conv3x3(drb, self.ws["bend.wpt"], None, F, out=g)
        for k in reversed(range(self.nb)):
            r_in, a = sv["blocks"][k]
            p = f"body.{k}.body."
            other, da = gs[k], das[k]
            # r_out = rs*(conv2(a)+b2) + r_in ;  a = relu(conv1(r_in)+b1)
            wgrad(g, a, p + "2.weight", p + "2.bias")
            if fuse_rb:          # both data gradients of the block in one launch (da leaves for the weight gradient)
                ops.resblock64_bwd(g, self.ws[f"b{k}.2.wpt"], self.ws[f"b{k}.0.wpt"], a, rs, da, other)
                wgrad(da, r_in, p + "0.weight", p + "0.bias")
                g = other
                continue
            ops.conv3x3(g, self.ws[f"b{k}.2.wpt"], None, F, out=da, epi=4, R=a)     # * (a > 0)
            if rs != 1.0:
                ops.axpby(da, da, 0.0, rs)
            wgrad(da, r_in, p + "0.weight", p + "0.bias")
            ops.conv3x3(da, self.ws[f"b{k}.0.wpt"], None, F, out=other, epi=2, R=g)  # + skip gradient
            g = other
        if batched:
            flush()
        ops.axpby(g, drb, 1.0, 1.0)                          # long skip: rb = conv(body) + f0
        side(lambda: ops.conv3x3_cin1_wgrad(sv["x"], g, G("head.0.weight"), G("head.0.bias")))
        if use_side:
            main.wait_stream(self.wstream)                   # every weight gradient is in before anything reads them
        if rs != 1.0:                # d(conv2 weights) = rs * (g (x) a): g was used unscaled above
            for k in range(self.nb):
                for nm in (f"body.{k}.body.2.weight", f"body.{k}.body.2.bias"):
                    ops.axpby(G(nm), G(nm), 0.0, rs)
        if need_dx:
            wflip = net.head[0].weight.data.flip(2, 3).reshape(1, F, 3, 3).contiguous()
            return ops.conv3x3_cout1_fwd(g, wflip, None)
        return None
