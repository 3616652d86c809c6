"""SwinIR configurations the fused engine (srhip/swinir_engine.py: 8 x 8 windows = 64 tokens, head_dim ** -0.5) does not take --
any window_size, qk_scale -- on the general tape graph (srhip/tape.py).  Reference dlib/models/network_swinir.py:287-337 (block),
:140-179 (WindowAttention), :907-970 (forward).  Every window attention runs on rows per (window, head): batched GEMMs around
the row softmax, the relative-position bias (+ the shift mask per window) as a periodic addend, its table's gradient gathered
back through the index; roll / window partition / reverse are the relayout copies around it; Linears on the exact-f32 GEMM,
3 x 3 convs on the conv kernels (the image-channel convs of RGB nets and the '3conv' residual convs as im2col GEMMs).
Slower than the fused engine by design (nothing is fused); the same kernels' parity."""
import math

import torch

from . import ops
from .swinir_engine import _Bufs
from .tape import Tape, WeightBank


class SwinIRTapeEngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.bufs = _Bufs()
        self.bank = WeightBank()
        self.prepared = False
        self.saved = None
        self._masks = {}

    def invalidate(self):
        self.prepared = False

    def bucket_prefixes(self):
        return [[""]]

    def _convs(self):
        net = self.net
        out = []
        conv3 = getattr(net, "resi_connection", "1conv") == "3conv"
        for li, layer in enumerate(net.layers):
            if not conv3:
                out.append((f"layers.{li}.conv", layer.conv))
        if not conv3:
            out.append(("conv_after_body", net.conv_after_body))
        if net.upsampler == "pixelshuffledirect":
            out.append(("upsample.0", net.upsample[0]))
        elif net.upsampler == "nearest_conv":
            out += [("conv_before_upsample.0", net.conv_before_upsample[0]), ("conv_up1", net.conv_up1), ("conv_up2", net.conv_up2),
                    ("conv_hr", net.conv_hr)]
        else:
            out.append(("conv_before_upsample.0", net.conv_before_upsample[0]))
            out += [(f"upsample.{2 * i}", net.upsample[2 * i]) for i in range(int(round(math.log2(net.upscale))))]
        return out

    def prepare(self):
        self.bank.begin()
        for key, m in self._convs():
            self.bank.conv(key, m.weight, m.bias, "c3")
        self.bank.finish(self.net.conv_first.weight.device)
        self.prepared = True

    def forward(self, x, dp=None, save=True):
        """x [B, H, W] (1 channel) or NHWC [B, H, W, 4] (in_chans 2..4 zero-padded, as SwinIR.prepare_input hands it over), H and
        W multiples of the window -> [B, in_chans, s H, s W]; dp: None or [2 nblocks, B] DropPath multipliers."""
        from dlib.models.network_swinir import _shift_mask
        if not self.prepared:
            self.prepare()
        net = self.net
        B, H, W = x.shape[:3]
        ci = net.in_chans
        C, L = net.embed_dim, H * W
        T = B * L
        dev = x.device
        t = Tape(self.bufs, self.bank, save, dev)
        nm = {id(p): k for k, p in net.named_parameters()}
        N = lambda p: None if p is None else nm[id(p)]

        def lin(v, m):
            return t.linear(v, m.weight, m.bias, N(m.weight), N(m.bias))

        def ln(v, m):
            return t.layernorm_rows(v, m, N(m.weight), N(m.bias), eps=1e-5)

        def c3(v, key, m, **kw):
            return t.conv(v, key, (N(m.weight), N(m.bias)), **kw)

        def resi(v, key, m, skip):
            """the conv in front of a residual connection ('1conv' | '3conv', :543-552) + skip"""
            if getattr(net, "resi_connection", "1conv") == "1conv":
                return c3(v, key, m, res=(skip, 1.0))
            # '3conv': conv C -> C/4, LeakyReLU(0.2), 1x1, LeakyReLU(0.2), conv C/4 -> C as im2col GEMMs (any channel count)
            c0 = t.relu(t.conv_im2col(v, m[0].weight, m[0].bias, N(m[0].weight), N(m[0].bias), 3), 0.2)
            c1 = t.relu(t.conv_im2col(c0, m[2].weight, m[2].bias, N(m[2].weight), N(m[2].bias), 1), 0.2)
            return t.axpby(t.conv_im2col(c1, m[4].weight, m[4].bias, N(m[4].weight), N(m[4].bias), 3), skip)

        def to_image(v):
            """NHWC [B, h, w, ci] -> the output layout [B, ci, h, w]"""
            return t.relayout(v, lambda u: u.permute(0, 3, 1, 2), lambda g: g.permute(0, 2, 3, 1))

        def last_conv(u):
            m = net.conv_last
            if ci == 1:
                return t.conv_out1(u, m.weight, m.bias, (N(m.weight), N(m.bias)))
            return to_image(t.conv_im2col(u, m.weight, m.bias, N(m.weight), N(m.bias), 3))

        def drop_path(v, k):
            if dp is None:
                return v
            return t.scale_rows(v, dp[k].repeat_interleave(L).contiguous(), lambda d: None)

        def block(blk, v, bi):
            ws, shift, heads = blk.window_size, blk.shift_size, blk.num_heads
            if H % ws or W % ws:
                raise NotImplementedError("SwinIR (tape graph): the block's window does not divide the padded input")
            dh = C // heads
            dh4 = (dh + 3) & ~3
            n, nW = ws * ws, (H // ws) * (W // ws)
            scale = float(getattr(net, "qk_scale", None) or dh ** -0.5)
            qkv = lin(ln(v, blk.norm1), blk.attn.qkv)
            parts = []
            for j in range(3):
                rows, fwd, inv = t.window_heads(t.reshape(t.cols(qkv, j * C, (j + 1) * C), B, H, W, C), (ws, ws), heads, shift, dh4)
                parts.append(rows)
            tab = blk.attn.relative_position_bias_table
            idx = blk.attn.relative_position_index.reshape(-1).to(dev)
            bias = tab.data[idx].view(n, n, heads).permute(2, 0, 1)                         # [heads, n, n]
            if shift:
                key = (H, W, ws, shift)
                if key not in self._masks:  # made once per input size (host-built: not inside a graph capture)
                    self._masks[key] = _shift_mask(H, W, ws, shift).to(dev)
                mask = self._masks[key]                                                      # [nW, n, n]
                addend = (bias.unsqueeze(0) + mask.unsqueeze(1)).reshape(-1, n, n).contiguous()
            else:
                addend = bias.contiguous()

            def on_dbias(d):
                db = d.reshape(-1, heads, n, n).sum(0)

                def prod(o):
                    o.zero_()
                    o.index_add_(0, idx, db.permute(1, 2, 0).reshape(-1, heads))
                t.gparam(N(tab), prod)
            o = t.attend(parts[0], parts[1], parts[2], B * nW * heads, n, n, 1, dh4, scale, bias=addend, on_dbias=on_dbias)
            a = t.reshape(t.heads_windows(o, fwd, inv, dh, dh4), T, C)
            x1 = t.axpby(drop_path(lin(a, blk.attn.proj), 2 * bi), v)
            m = lin(t.unary(lin(ln(x1, blk.norm2), blk.mlp.fc1), "gelu"), blk.mlp.fc2)
            return t.axpby(drop_path(m, 2 * bi + 1), x1)

        cf = net.conv_first
        if ci == 1:
            f0 = t.conv_in1(x, cf.weight, cf.bias, (N(cf.weight), N(cf.bias)))
        else:
            f0 = t.conv_im2col(t.var(x[..., :ci].contiguous(), need=False), cf.weight, cf.bias, N(cf.weight), N(cf.bias), 3)
        tk = t.reshape(f0, T, C)
        if net.patch_norm:
            tk = ln(tk, net.patch_embed.norm)
        if net.ape:
            tk = t.add_rows_param(tk, net.absolute_pos_embed, N(net.absolute_pos_embed))
        bi = 0
        for li, layer in enumerate(net.layers):
            t_in = tk
            for blk in layer.residual_group.blocks:
                tk = block(blk, tk, bi)
                bi += 1
            tk = t.reshape(resi(t.reshape(tk, B, H, W, C), f"layers.{li}.conv", layer.conv, t.reshape(t_in, B, H, W, C)), T, C)
        tk = ln(tk, net.norm)
        f = resi(t.reshape(tk, B, H, W, C), "conv_after_body", net.conv_after_body, f0)
        s = net.upscale
        if net.upsampler == "pixelshuffledirect":
            y = t.shuffle(c3(f, "upsample.0", net.upsample[0]), s)
            y = t.reshape(y, B, H * s, W * s) if ci == 1 else to_image(y)
        elif net.upsampler == "nearest_conv":
            u = t.relu(c3(f, "conv_before_upsample.0", net.conv_before_upsample[0]), 0.01)
            u = t.relu(c3(t.nearest_up2(u), "conv_up1", net.conv_up1), 0.2)
            u = t.relu(c3(t.nearest_up2(u), "conv_up2", net.conv_up2), 0.2)
            y = last_conv(t.relu(c3(u, "conv_hr", net.conv_hr), 0.2))
        else:
            u = t.relu(c3(f, "conv_before_upsample.0", net.conv_before_upsample[0]), 0.01)
            for i in range(int(round(math.log2(s)))):
                u = t.shuffle(c3(u, f"upsample.{2 * i}", net.upsample[2 * i]), 2)
            y = last_conv(u)
        if save:
            self.saved = (t, y)
        return y.t.reshape(B, ci, H * s, W * s)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        assert self.saved is not None, "backward() without a saved forward"
        if need_dx:
            raise NotImplementedError("SwinIR (tape graph): no gradient with respect to the input image")
        tape, out = self.saved
        tape.backward(out, dy.reshape(out.t.shape).contiguous(), grads)
        return None
