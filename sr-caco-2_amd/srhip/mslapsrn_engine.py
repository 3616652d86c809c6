"""MSLapSRN forward / backward as a fixed sequence of libsrhip launches (the plain CNNs reuse the
3x3-conv kernels).

Reference: dlib/models/network_mslapsr.py:67-174 -- conv 1->64 + LeakyReLU(0.2); per octave (x2 each): ten
(conv 64->64 + LeakyReLU), ConvTranspose2d(64, 64, 4, 2, 1) + LeakyReLU -> the octave's features; conv 64->1 on them +
ConvTranspose2d(1, 1, 4, 2, 1) of the current image -> the octave's image.  The last image is the output, the earlier
ones are ``intermediate_outs`` (the trainer adds their losses, model_plain.py:277-314).

No transposed-conv kernel is needed: ConvTranspose2d(k 4, stride 2, padding 1) == PixelShuffle(2) o Conv2d(3x3) with
four zero-padded 2x2 sub-kernels -- output pixel (2y+i, 2x+j) takes input pixels (y+dy, x+dx) with kernel tap
(i+1-2dy, j+1-2dx) where that lies inside the 4x4 kernel.  So the 64-channel transposed conv is the fused
conv + PixelShuffle kernel of the EDSR upsampler (ops.conv3x3_ps2*, LeakyReLU as its epilogue) on a derived dense
weight, the 1-channel one is the 1-channel edge conv with four outputs + the shuffle index kernel, and their weight
gradients are gathered back from the gradients of the dense forms (each of the 16 kernel taps is read by exactly one
(sub-pixel, tap) pair).  The LeakyReLU masks of the backward ride in the data-gradient convs (epi 7).  NHWC throughout.
"""
import math

import torch

from . import ops
from .swinir_engine import _Bufs

CH = 64
SLOPE = 0.2


def _convT_maps(ci, co, device):
    """Index maps between a ConvTranspose2d weight wT [ci, co, 4, 4] and its dense 3x3 form wc [4*co, ci, 3, 3]
    (PixelShuffle channel order co*4 + 2i + j): src[p] = flat index into wT read by wc.flat[p] (or -1),
    pos[q] = flat index into wc of the one position that reads wT.flat[q]."""
    src = torch.full((co, 4, ci, 3, 3), -1, dtype=torch.long)
    ids = torch.arange(ci * co * 16).view(ci, co, 4, 4)
    for i in range(2):
        for j in range(2):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    ky, kx = i + 1 - 2 * dy, j + 1 - 2 * dx
                    if 0 <= ky < 4 and 0 <= kx < 4:
                        src[:, 2 * i + j, :, dy + 1, dx + 1] = ids[:, :, ky, kx].t()
    src = src.reshape(-1)
    pos = torch.empty(ci * co * 16, dtype=torch.long)
    valid = src >= 0
    pos[src[valid]] = torch.nonzero(valid)[:, 0]
    return src.clamp_min(0).to(device), valid.to(device), pos.to(device)


class MSLapSRNEngine:
    def __init__(self, net):
        self.net = net
        self.octaves = int(math.log2(net.upscale))
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()
        if not (self.ws.use_bx3 and ops.bx3_nt_for(CH) and ops.ps2_fusable(CH, 4 * CH)):
            raise NotImplementedError("MSLapSRN (libsrhip) runs on the bf16x3 kernels (SRHIP_MM=f32 is not supported)")
        self._prep = self._prep_sig = None
        self._maps = None
        self.prepared = False
        self.saved = None
        self.intermediate_outs = []

    def invalidate(self):
        self.prepared = False

    def bucket_prefixes(self):
        """0.4-1.2 M parameters: one gradient bucket."""
        return [["conv1.", "laplacian_pyramid_conv"]]

    # names of the reference modules of octave o (network_mslapsr.py:85-133)
    def _names(self, o):
        return tuple(f"laplacian_pyramid_conv{3 * o + k}" for k in (1, 2, 3))

    def _mods(self, o):
        net = self.net
        return tuple(getattr(net, n) for n in self._names(o))

    def prepare(self):
        D, ws = self.derived, self.ws
        dev = self.net.conv1[0].weight.device
        if self._maps is None:
            self._maps = (_convT_maps(CH, CH, dev), _convT_maps(1, 1, dev))
        (srcF, validF, _), (src1, valid1, _) = self._maps
        sig = tuple(p.data_ptr() for p in self.net.parameters())
        rebuild = self._prep is None or sig != self._prep_sig
        tb = ops.PrepTable() if rebuild else None
        for o in range(self.octaves):
            a, b, _ = self._mods(o)
            # dense 3x3 forms of the two transposed convs (+ biases repeated over the 4 sub-pixels)
            wc = D.get(f"o{o}.wc", 4 * CH, CH, 3, 3, device=dev)
            wc.view(-1).copy_(torch.where(validF, a[10].weight.data.reshape(-1)[srcF], wc.new_zeros(())))
            D.get(f"o{o}.bc", 4 * CH, device=dev).copy_(a[10].bias.data.repeat_interleave(4))
            w4 = D.get(f"o{o}.w4", 4, 1, 3, 3, device=dev)
            w4.view(-1).copy_(torch.where(valid1, b.weight.data.reshape(-1)[src1], w4.new_zeros(())))
            D.get(f"o{o}.b4", 4, device=dev).copy_(b.bias.data.repeat_interleave(4))
            D.get(f"o{o}.w4f", 1, 4, 3, 3, device=dev).copy_(w4.flip(2, 3).reshape(1, 4, 3, 3))
            if rebuild:
                for k in range(10):
                    w = a[k].cl[0].weight.data
                    tb.conv(w, ws.planes(f"o{o}.c{k}.wp", 9 * CH, CH, dev))
                    tb.conv(w, ws.planes(f"o{o}.c{k}.wpt", 9 * CH, CH, dev), data_grad=True)
                tb.conv(wc, ws.planes(f"o{o}.up.wp", 9 * 4 * CH, CH, dev), ps2=True)
                tb.conv(wc, ws.planes(f"o{o}.up.wpt", 9 * CH, 4 * CH, dev), data_grad=True, ps2=True)
        if rebuild:
            self._prep, self._prep_sig = tb.build(dev), sig
        self._prep.run()
        self.prepared = True

    def forward_h16(self, x):
        """--amp evaluation on fp16 storage (conv_h16.hip): the feature branch on float16 maps, one fp16 product; the
        1-channel image branch stays f32."""
        net, D, ws = self.net, self.derived, self.ws
        B, h, w = x.shape
        dev = x.device

        def buf(name, *shape):
            return self.bufs.get("h." + name, *shape, device=dev, dtype=torch.float16)
        feat = ops.conv3x3_cin1_h16(x, net.conv1[0].weight.data, net.conv1[0].bias.data, CH, out=buf("f0", B, h, w, CH), leaky=SLOPE)
        img, outs = x, []
        for o in range(self.octaves):
            a, _, c = self._mods(o)
            for k in range(10):
                feat = ops.conv3x3_h16(feat, ws[f"o{o}.c{k}.wp"], a[k].cl[0].bias.data, CH, out=buf(f"o{o}.a{k % 2}", B, h, w, CH),
                                       epi=6, alpha=SLOPE)
            up = ops.conv3x3_h16(feat, ws[f"o{o}.up.wp"], D.d[f"o{o}.bc"], 4 * CH, out=buf(f"o{o}.up", B, 2 * h, 2 * w, CH), epi=6,
                                 alpha=SLOPE, ps2=True)
            c4 = self.bufs.get(f"h.o{o}.c4", B, h, w, 4, device=dev)
            ops.conv3x3_cin1_fwd(img, D.d[f"o{o}.w4"], D.d[f"o{o}.b4"], 4, out=c4)
            base = self.bufs.get(f"h.o{o}.base", B, 1, 2 * h, 2 * w, device=dev)
            ops.pixel_shuffle(c4, 2, out=base)
            out = torch.empty(B, 2 * h, 2 * w, device=dev)
            ops.conv3x3_cout1_h16(up, c.weight.data, c.bias.data, add=base.view(B, 2 * h, 2 * w), out=out)
            outs.append(out.view(B, 1, 2 * h, 2 * w))
            feat, img, h, w = up, out, 2 * h, 2 * w
        self.intermediate_outs = outs[:-1]
        return outs[-1]

    def _h16_ok(self):
        return all(self.ws[f"o{o}.c{k}.wp"].fmt == 1 for o in range(self.octaves) for k in range(10)) and \
            all(self.ws[f"o{o}.up.wp"].fmt == 1 for o in range(self.octaves))

    # ------------------------------------------------------------------ forward
    def forward(self, x, dp=None, save=True):
        """x [B,H,W] (LR) -> [B,1,s*H,s*W]; the images of the earlier octaves in self.intermediate_outs."""
        if not self.prepared:
            self.prepare()
        if not save and ops.h16_eval() and self._h16_ok():
            self.last_eval_path = "fp16 storage"
            return self.forward_h16(x)
        net, D, ws = self.net, self.derived, self.ws
        B, h, w = x.shape
        dev = x.device
        tag = "t" if save else "e"

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        feat = buf("f0", B, h, w, CH)
        ops.conv3x3_cin1_fwd(x, net.conv1[0].weight.data, net.conv1[0].bias.data, CH, out=feat)
        ops.leaky_relu_(feat, SLOPE)
        img = x
        outs, sv_oct = [], []
        for o in range(self.octaves):
            a, _, c = self._mods(o)
            acts = [feat]
            for k in range(10):
                nxt = buf(f"o{o}.a{k + 1 if save else 1 + k % 2}", B, h, w, CH)
                ops.conv3x3(feat, ws[f"o{o}.c{k}.wp"], a[k].cl[0].bias.data, CH, out=nxt, epi=6, alpha=SLOPE)
                if save:
                    acts.append(nxt)
                feat = nxt
            up = buf(f"o{o}.up", B, 2 * h, 2 * w, CH)
            ops.conv3x3_ps2(feat, ws[f"o{o}.up.wp"], D.d[f"o{o}.bc"], up, epi=6, alpha=SLOPE)
            last = o + 1 == self.octaves
            out = buf(f"o{o}.img", B, 1, 2 * h, 2 * w) if (save or not last) else torch.empty(B, 1, 2 * h, 2 * w, device=dev)
            c4 = buf(f"o{o}.c4", B, h, w, 4)
            ops.conv3x3_cin1_fwd(img, D.d[f"o{o}.w4"], D.d[f"o{o}.b4"], 4, out=c4)
            ops.pixel_shuffle(c4, 2, out=out)
            res = buf(f"o{o}.res", B, 2 * h, 2 * w)
            ops.conv3x3_cout1_fwd(up, c.weight.data, c.bias.data, out=res)
            ops.axpby(out.view(B, 2 * h, 2 * w), res, 1.0, 1.0)
            if save:
                sv_oct.append(dict(acts=acts, up=up, img_in=img, h=h, w=w))
            outs.append(out)
            feat, img, h, w = up, out.view(B, 2 * h, 2 * w), 2 * h, 2 * w
        self.intermediate_outs = outs[:-1]
        if save:
            self.bufs.d["t.y"] = outs[-1]          # where ModelPlain looks for the step's output
            self.saved = dict(x=x, B=B, oct=sv_oct)
        return outs[-1]

    # ------------------------------------------------------------------ backward
    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False, d_inter=None):
        """dy: gradient of the output [B,1,sH,sW]; d_inter: gradients of intermediate_outs (list, entries may be
        None) -- the trainer's multi-scale loss (model_plain.py:277-314)."""
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        assert not need_dx, "MSLapSRN (libsrhip): no gradient with respect to the input image"
        net, D, ws = self.net, self.derived, self.ws
        B = sv["B"]
        dev = dy.device
        (_, _, posF), (_, _, pos1) = self._maps

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        g_img = None        # gradient wrt the octave's image arriving from the NEXT octave's image branch
        g_feat = None       # gradient wrt the octave's features arriving from the NEXT octave's first conv
        for o in reversed(range(self.octaves)):
            an, bn, cn = self._names(o)
            a, b, c = self._mods(o)
            so = sv["oct"][o]
            h, w, acts, up = so["h"], so["w"], so["acts"], so["up"]
            d_loss = dy if o + 1 == self.octaves else (None if d_inter is None else d_inter[o])
            d_out = buf(f"o{o}.dout", B, 2 * h, 2 * w)
            if d_loss is not None:
                d_out.copy_(d_loss.reshape(B, 2 * h, 2 * w))
                if g_img is not None:
                    ops.axpby(d_out, g_img, 1.0, 1.0)
            else:
                assert g_img is not None
                d_out.copy_(g_img)
            # ---- conv 64 -> 1 on the features (1-channel kernels with x / dy swapped and flipped taps)
            ops.conv3x3_cin1_wgrad(d_out, up, grads[cn + ".weight"], None, flip=True)
            ops.sum_into(d_out, grads[cn + ".bias"])
            g_up = buf(f"o{o}.gup", B, 2 * h, 2 * w, CH)
            ops.conv3x3_cin1_fwd(d_out, c.weight.data, None, CH, out=g_up, flip=True)
            if g_feat is not None:
                ops.axpby(g_up, g_feat, 1.0, 1.0)
            ops.leaky_relu_mask(g_up, up, SLOPE)
            # ---- transposed conv 64 -> 64: dense-form gradients, gathered back to the [ci, co, 4, 4] layout
            dwc, dbc = buf("dwc", 4 * CH, CH, 3, 3), buf("dbc", 4 * CH)
            ops.conv3x3_wgrad(g_up, acts[10], dwc, dbc, ps2=True)
            grads[an + ".10.weight"].view(-1).copy_(dwc.view(-1)[posF])
            grads[an + ".10.bias"].copy_(dbc.view(CH, 4).sum(1))
            gs = [buf(f"o{o}.g{k}", B, h, w, CH) for k in range(10)]     # d / d(pre-activation of conv k)
            ops.conv3x3_ps2_bwd_data(g_up, ws[f"o{o}.up.wpt"], gs[9], epi=7, R=acts[10], alpha=SLOPE)
            wg = []
            for k in reversed(range(10)):
                wg.append((gs[k], acts[k], grads[f"{an}.{k}.cl.0.weight"], grads[f"{an}.{k}.cl.0.bias"]))
                if k > 0:
                    ops.conv3x3(gs[k], ws[f"o{o}.c{k}.wpt"], None, CH, out=gs[k - 1], epi=7, R=acts[k], alpha=SLOPE)
            gin = buf(f"o{o}.gin", B, h, w, CH)                           # d / d(octave input features)
            if o > 0:       # = the previous octave's `up`: masked there, after the conv 64->1 branch joined
                ops.conv3x3(gs[0], ws[f"o{o}.c0.wpt"], None, CH, out=gin)
            else:           # = LeakyReLU(conv1(x))
                ops.conv3x3(gs[0], ws[f"o{o}.c0.wpt"], None, CH, out=gin, epi=7, R=acts[0], alpha=SLOPE)
            ops.conv3x3_wgrad_batched(wg)
            g_feat = gin
            # ---- image branch: transposed conv 1 -> 1 = 1-channel conv with four outputs + shuffle
            d4 = buf(f"o{o}.d4", B, h, w, 4)
            ops.pixel_shuffle(d_out.view(B, 1, 2 * h, 2 * w), 2, inverse=True, out=d4)
            dw4, db4 = buf("dw4", 4, 1, 3, 3), buf("db4", 4)
            ops.conv3x3_cin1_wgrad(so["img_in"], d4, dw4, db4)
            grads[bn + ".weight"].view(-1).copy_(dw4.view(-1)[pos1])
            grads[bn + ".bias"].copy_(db4.sum().reshape(1))
            if o > 0:
                g_img = buf(f"o{o}.gimg", B, h, w)
                ops.conv3x3_cout1_fwd(d4, D.d[f"o{o}.w4f"], None, out=g_img)
        ops.conv3x3_cin1_wgrad(sv["x"], g_feat, grads["conv1.0.weight"], grads["conv1.0.bias"])
        return None
