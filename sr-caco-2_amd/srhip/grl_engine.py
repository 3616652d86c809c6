"""GRL on libsrhip (reference dlib/models/network_grl.py:1462-1512): stages of mixed-attention blocks
(:1061-1076) -- one qkv Linear whose halves feed an 8x8 (shifted) window attention and an anchored stripe attention
(average-pooled 4x4 anchors attend to the 8x8 stripe, the stripe attends back to the anchors), both cosine attentions with a
clamped logit scale and a 16 sigmoid(CPB MLP) bias, a shared output projection, post-norm residuals, a conv + GELU + conv +
channel-attention local branch, a GELU MLP -- each stage closed by a 3x3 conv and a skip, and a pixel-shuffle tail.  Written
directly over the libsrhip ops: Linears and 3x3 convs on the GEMM / conv kernels (fp16x2 / bf16x3 planes from 64 channels on
-- the kernels --amp narrows to one product -- exact f32 below; GELU as the fc2 prologue; the
C/4 channels of the local branch zero-padded to a multiple of 4: exact), attentions / pooling / bias images in grl_ops.hip,
the channel gate and LayerNorms of the earlier nets.  Tokens stay channels-last [B, H, W, C] throughout, so blc<->bchw,
roll, window_partition and window_reverse are address arithmetic inside the kernels.  Training (save=True) runs the graph on the tape of srhip/tape.py (_forward_tape)."""
import torch
import torch.nn.functional as F

from . import ops
from .planes import PlaneCache
from .swinir_engine import _Bufs
from .tape import Tape, WeightBank


def _pad4(n):
    return (n + 3) // 4 * 4


class GRLEngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.prepared = True
        self.saved = None
        self.taps = None
        self._w = {}
        self._consts = {}                 # input-size constants (shift masks): survive invalidate()
        self.planes = PlaneCache()
        self.bufs = _Bufs()
        self.bank = WeightBank()
        self._bank_ready = False

    def invalidate(self):
        self._w = {}
        self.planes.clear()
        self._bank_ready = False

    def bucket_prefixes(self):
        return [[""]]

    # ------------------------------------------------------------------ derived weights (cached until the weights change)
    def _pack(self, conv, cin_pad=None, cout_pad=None):
        return self.planes.conv(id(conv), conv.weight.data, conv.bias.data, cin_pad, cout_pad)

    def _lin(self, x, m, **kw):
        w, b = self.planes.linear(id(m), m.weight.data, None if m.bias is None else m.bias.data)
        return ops.gemm_nt(x, w, b, **kw)

    def _conv3(self, x, conv, **kw):
        wp, b, co = self._pack(conv)
        return ops.conv3x3(x, wp, b, co, **kw)

    def _bias_image(self, tr, table, index, heads):
        """AffineTransform (:305-311): the CPB MLP over the relative-coordinates table, gathered by the index, 16 sigmoid(.)"""
        key = ("bias", id(tr), tuple(index.shape))
        if key not in self._w:
            dev = tr.logit_scale.device
            t = table.reshape(-1, 2).to(dev)
            a = torch.zeros(t.shape[0], 4, device=dev)
            a[:, :2] = t
            w0 = torch.zeros(512, 4, device=dev)
            w0[:, :2] = tr.cpb_mlp[0].weight.data
            h = ops.gemm_nt(a, w0, tr.cpb_mlp[0].bias.data, epi=1)
            w2 = torch.zeros(_pad4(heads), 512, device=dev)
            w2[:heads] = tr.cpb_mlp[2].weight.data
            tab = ops.gemm_nt(h, w2)[:, :heads].contiguous()
            self._w[key] = (ops.cpb_bias(tab, index.to(dev).contiguous(), heads),
                            tr.logit_scale.data.reshape(-1).contiguous())
        return self._w[key]

    @staticmethod
    def _ln_res(y, res, norm):
        """y <- res + norm(y): a post-norm residual"""
        if y.shape[1] <= 256:
            return ops.layernorm_rows_res(y, res, norm.weight.data, norm.bias.data, y)
        ops.layernorm_rows(y, norm.weight.data, norm.bias.data, y)
        ops.axpby(y, res, 1.0, 1.0)
        return y

    # ------------------------------------------------------------------ one block (x [T, C], channels-last tokens)
    def _block(self, blk, x, B, H, W, i, s, name):
        net = self.net
        C = net.embed_dim
        half = C // 2
        df = net.df
        a = blk.attn
        ws = net.window_size
        ssz = net.stripe_size if i % 2 == 0 else net.stripe_size[::-1]       # 'H' / 'W' stripes (:141-143, :1012-1017)
        asz = [v // df for v in ssz]
        sfx = "h" if i % 2 == 0 else "v"
        shift = ws[0] // 2 if i % 2 == 0 else 0
        hw, hs = net.heads_w[s], net.heads_s[s]
        x4 = x.view(B, H, W, C)

        qkv = self._lin(x, a.qkv.body).view(B, H, W, 3 * C)
        anchor = self._lin(ops.avgpool2d(x4, df).view(-1, C), a.anchor.body[0].reduction).view(B, H // df, W // df, half)
        att = torch.empty(B, H, W, C, device=x.device)
        bw, lw = self._bias_image(a.window_attn.attn_transform, net.table_w, net.index_w, hw)
        ops.cosine_window_attention(qkv[..., 0:half], ws, qkv[..., half:2 * half], qkv[..., 2 * half:3 * half], ws, lw, bw,
                                    att[..., :half], hw, half // hw, shift)
        o = 3 * half
        b1, l1 = self._bias_image(a.stripe_attn.attn_transform1, getattr(net, "table_s" + sfx),
                                  getattr(net, f"index_s{sfx}_a2w"), hs)
        b2, l2 = self._bias_image(a.stripe_attn.attn_transform2, getattr(net, "table_s" + sfx),
                                  getattr(net, f"index_s{sfx}_w2a"), hs)
        xa = torch.empty_like(anchor)
        ops.cosine_window_attention(anchor, asz, qkv[..., o + half:o + 2 * half], qkv[..., o + 2 * half:o + 3 * half], ssz, l1, b1,
                                    xa, hs, half // hs, 0)
        ops.cosine_window_attention(qkv[..., o:o + half], ssz, anchor, xa, asz, l2, b2, att[..., half:], hs, half // hs, 0)
        p = self._lin(att.view(-1, C), a.proj)
        if self.taps is not None:
            self.taps[name + ".attn"] = p.detach().clone().view(B, H * W, C)
        self._ln_res(p, x, blk.norm1)
        if net.local_connection:
            cab = blk.conv.cab
            cm = cab[0].weight.shape[0]
            # C/4 channels zero-padded (exact: gelu(0) = 0): to 64 where that moves both convs onto the bf16-plane kernels
            # (45 -> 64 at the registry's width, as SwinIR's '3conv'), else to a multiple of 4 for the exact-f32 ones
            cm = 64 if ops.bx3_nt_for(C) and 32 < cm <= 64 else _pad4(cm)
            wp1, bb1, _ = self._pack(cab[0], cout_pad=cm)
            c1 = ops.conv3x3(x4, wp1, bb1, cm)
            ops.unary(c1, c1, "gelu")
            wp2, bb2, _ = self._pack(cab[2], cin_pad=cm)
            c2 = ops.conv3x3(c1, wp2, bb2, C)
            ca = cab[3].attention
            xn = torch.empty_like(x)
            ops.channel_gate(c2, ca[1].weight.data.view(-1, C), ca[1].bias.data, ca[3].weight.data.view(C, -1), ca[3].bias.data,
                             p.view(B, H, W, C), c2, xn.view(B, H, W, C))
        else:
            xn = p
        h = self._lin(xn, blk.mlp.fc1)
        m = self._lin(h, blk.mlp.fc2, a_mode=2)
        self._ln_res(m, xn, blk.norm2)
        return m

    # ------------------------------------------------------------------ forward
    def forward(self, x3, dp=None, save=False):
        if save:
            if dp is None:          # the autograd path (TrainStep draws the multipliers itself)
                dp = self.net.sample_drop_path(x3.shape[0], x3.device)
            return self._forward_tape(x3, dp)
        net = self.net
        B, H0, W0 = x3.shape
        p = net.pad_size
        ph, pw = (p - H0 % p) % p, (p - W0 % p) % p
        if ph or pw:                                                          # check_image_size (:1415-1424)
            x3 = F.pad(x3[:, None], (0, pw, 0, ph), 'reflect' if ph < H0 and pw < W0 else 'constant')[:, 0]
        x3 = x3.contiguous()
        H, W = x3.shape[1:]
        per = H * W * 3 * net.embed_dim                                       # the GEMM kernels index 2^29 elements per operand
        nb = max(1, ((1 << 29) - 1) // per)
        y = torch.cat([self._forward(x3[b0:b0 + nb]) for b0 in range(0, B, nb)]) if B > nb else self._forward(x3)
        s = net.upscale
        return y[:, None, :H0 * s, :W0 * s].contiguous()

    def _forward(self, x3):
        net = self.net
        B, H, W = x3.shape
        C = net.embed_dim

        def tap(name, v):
            if self.taps is not None:
                self.taps[name] = v.detach().clone().view(B, H * W, -1)
        f0 = ops.conv3x3_cin1_fwd(x3, net.conv_first.weight.data, net.conv_first.bias.data, C)
        t = ops.layernorm_rows(f0.view(-1, C), net.norm_start.weight.data, net.norm_start.bias.data, torch.empty(B * H * W, C, device=x3.device))
        tap("start", t)
        for s, stage in enumerate(net.layers):
            res = t
            for i, blk in enumerate(stage.blocks):
                res = self._block(blk, res, B, H, W, i, s, f"layers.{s}.blocks.{i}")
                tap(f"layers.{s}.blocks.{i}", res)
            t = self._conv3(res.view(B, H, W, C), stage.conv, epi=2, R=t.view(B, H, W, C)).view(-1, C)
            tap(f"layers.{s}", t)
        ops.layernorm_rows(t, net.norm_end.weight.data, net.norm_end.bias.data, t)
        f = self._conv3(t.view(B, H, W, C), net.conv_after_body, epi=2, R=f0)
        tap("body", f)
        u = self._conv3(f, net.conv_before_upsample[0], epi=6, alpha=0.01)
        for conv in list(net.upsample.up)[0::2]:
            u = ops.pixel_shuffle(self._conv3(u, conv), 2, nhwc_out=True)
        return ops.conv3x3_cout1_fwd(u, net.conv_last.weight.data, net.conv_last.bias.data)

    # ------------------------------------------------------------------ training: the same graph on the tape
    def _forward_tape(self, x3, dp=None):
        """forward() with the tape recording (srhip/tape.py).  Every attention runs on rows per (window, head) -- L2-normalised
        queries / keys, the clamped logit scale on the query rows, 16 sigmoid(CPB MLP)[index] (+ the shift mask) as a
        periodic addend, batched GEMMs around the row softmax; roll / window partition / reverse are the relayout copies
        around them; the C/4-channel convs of the local branch as im2col + GEMM; the CPB MLPs (a [225, 2] table through 512
        units) by hand."""
        import math
        from dlib.models.network_grl import table_index_mask
        net = self.net
        B, H, W = x3.shape
        C, df, ws = net.embed_dim, net.df, list(net.window_size)
        half = C // 2
        if H % net.pad_size or W % net.pad_size:
            raise NotImplementedError("GRL on libsrhip: training patches have to be multiples of the window / stripe size")
        dev = x3.device
        if not self._bank_ready:
            self.bank.begin()
            for si, stage in enumerate(net.layers):
                self.bank.conv(f"layers.{si}.conv", stage.conv.weight, stage.conv.bias, "c3")
            self.bank.conv("conv_after_body", net.conv_after_body.weight, net.conv_after_body.bias, "c3")
            self.bank.conv("conv_before_upsample.0", net.conv_before_upsample[0].weight, net.conv_before_upsample[0].bias, "c3")
            for k, conv in enumerate(list(net.upsample.up)[0::2]):
                self.bank.conv(f"upsample.up.{2 * k}", conv.weight, conv.bias, "c3")
            self.bank.finish(dev)
            self._bank_ready = True
        t = Tape(self.bufs, self.bank, True, dev)
        nm = {id(p): k for k, p in net.named_parameters()}
        N = lambda p: None if p is None else nm[id(p)]
        T = B * H * W
        key = ("mask_w", H, W)
        if key not in self._consts:         # made once per input size (a host-built table: not inside a graph capture)
            self._consts[key] = table_index_mask((H, W), net.window_size, net.stripe_size, df)["mask_w"].to(dev)
        mask_w = self._consts[key]                                                                      # [nW, 64, 64]

        def lin(x, m):
            return t.linear(x, m.weight, m.bias, N(m.weight), N(m.bias))

        def ln(x, m):
            return t.layernorm_rows(x, m, N(m.weight), N(m.bias))

        def per_head(m, Hm, Wm, wsz, heads, shift, pad):
            return t.window_heads(m, wsz, heads, shift, pad)

        def cosine_attention(tr, qr, kr, vr, nWin, heads, Tq, Tk, table, index, mask):
            """qr / kr / vr: per-head rows; returns per-head rows [(b, window, head, Tq), dh4]"""
            lam = tr.logit_scale
            sc = torch.clamp(lam.data.reshape(-1), max=math.log(100.0)).exp()                        # [heads]
            s_rows = sc.view(1, heads, 1).expand(B * nWin, heads, Tq).reshape(-1).contiguous()

            def on_scale(ds):
                d = ds.view(B * nWin, heads, Tq).sum((0, 2)) * sc * (lam.data.reshape(-1) < math.log(100.0))
                t.gparam(N(lam), lambda o: o.view(-1).copy_(d))
            # the CPB MLP over the relative-coordinates table (AffineTransform :296-319), by hand: [n, 2] -> 512 -> heads
            w0, b0, w2 = tr.cpb_mlp[0].weight, tr.cpb_mlp[0].bias, tr.cpb_mlp[2].weight
            tab = table.reshape(-1, 2).to(dev)
            pre = ops.mm(tab, w0.data, tb=True, bias=b0.data)
            h1 = torch.relu(pre)
            sg = torch.sigmoid(ops.mm(h1, w2.data, tb=True))                                                     # [n, heads]
            idx = index.reshape(-1).to(dev)
            bias = (16.0 * sg)[idx].view(Tq, Tk, heads).permute(2, 0, 1)                             # [heads, Tq, Tk]
            if mask is None:
                addend = bias.contiguous()
            else:
                addend = (bias.unsqueeze(0) + mask.unsqueeze(1)).reshape(-1, Tq, Tk).contiguous()    # [(window, head), Tq, Tk]

            def on_dbias(d):
                db = d.reshape(-1, heads, Tq, Tk).sum(0)                                             # [heads, Tq, Tk]
                dsg = torch.zeros_like(sg).index_add_(0, idx, db.permute(1, 2, 0).reshape(-1, heads)) * 16.0
                dz = dsg * sg * (1.0 - sg)
                t.gparam(N(w2), lambda o: o.copy_(ops.mm(dz, h1, ta=True)))
                dh1 = ops.mm(dz, w2.data) * (pre > 0)
                t.gparam(N(w0), lambda o: o.copy_(ops.mm(dh1, tab, ta=True)))
                t.gparam(N(b0), lambda o: o.copy_(dh1.sum(0)))
            qs = t.scale_rows(t.normalize_rows(qr), s_rows, on_scale)
            return t.attend(qs, t.normalize_rows(kr), vr, B * nWin * heads, Tq, Tk, 1, qr.t.shape[1], 1.0, bias=addend,
                            on_dbias=on_dbias)

        def drop(v, k):
            """timm DropPath with the multipliers of branch k (network_grl.py:1058-1066)"""
            if dp is None:
                return v
            return t.scale_rows(v, dp[k].repeat_interleave(H * W).contiguous(), lambda d: None)

        def avgpool(xmap):
            """nn.AvgPool2d(df): the df x df patches (F.unfold) against a constant averaging row"""
            Cc = xmap.t.shape[3]
            u = t.unfold(xmap, df, df)                                                                # [B nT, Cc df df]
            wavg = torch.zeros(4, df * df, device=dev)
            wavg[0] = 1.0 / (df * df)
            r = t.linear(t.reshape(u, -1, df * df), wavg, None, None)
            return t.reshape(t.cols(r, 0, 1), B, H // df, W // df, Cc)

        def block(blk, x, i, si, name):
            a = blk.attn
            hw, hs = net.heads_w[si], net.heads_s[si]
            ssz = list(net.stripe_size) if i % 2 == 0 else list(net.stripe_size)[::-1]
            asz = [v // df for v in ssz]
            sfx = "h" if i % 2 == 0 else "v"
            shift = ws[0] // 2 if i % 2 == 0 else 0
            xmap = t.reshape(x, B, H, W, C)
            qkv = lin(x, a.qkv.body)
            anchor = t.reshape(lin(t.reshape(avgpool(xmap), -1, C), a.anchor.body[0].reduction), B, H // df, W // df, half)

            def part(j):
                return t.reshape(t.cols(qkv, j * half, (j + 1) * half), B, H, W, half)
            # window attention
            dw4 = (half // hw + 3) & ~3
            nW = (H // ws[0]) * (W // ws[1])
            q, fwd_w, inv_w = per_head(part(0), H, W, ws, hw, shift, dw4)
            k, _, _ = per_head(part(1), H, W, ws, hw, shift, dw4)
            v, _, _ = per_head(part(2), H, W, ws, hw, shift, dw4)
            tr = a.window_attn.attn_transform
            ow = cosine_attention(tr, q, k, v, nW, hw, ws[0] * ws[1], ws[0] * ws[1], net.table_w, net.index_w,
                                  mask_w if shift else None)
            dhw = half // hw
            xw = t.heads_windows(ow, fwd_w, inv_w, dhw, dw4)
            # anchored stripe attention
            ds4 = (half // hs + 3) & ~3
            dhs = half // hs
            nS = (H // ssz[0]) * (W // ssz[1])
            q, fwd_s, inv_s = per_head(part(3), H, W, ssz, hs, 0, ds4)
            k, _, _ = per_head(part(4), H, W, ssz, hs, 0, ds4)
            v, _, _ = per_head(part(5), H, W, ssz, hs, 0, ds4)
            an, _, _ = per_head(anchor, H // df, W // df, asz, hs, 0, ds4)
            Ta, Ts = asz[0] * asz[1], ssz[0] * ssz[1]
            st = a.stripe_attn
            xa = cosine_attention(st.attn_transform1, an, k, v, nS, hs, Ta, Ts, getattr(net, "table_s" + sfx),
                                  getattr(net, f"index_s{sfx}_a2w"), None)
            os_ = cosine_attention(st.attn_transform2, q, an, xa, nS, hs, Ts, Ta, getattr(net, "table_s" + sfx),
                                   getattr(net, f"index_s{sfx}_w2a"), None)
            xs = t.heads_windows(os_, fwd_s, inv_s, dhs, ds4)
            return xw, xs, xmap, a

        f0 = t.conv_in1(x3, net.conv_first.weight, net.conv_first.bias, (N(net.conv_first.weight), N(net.conv_first.bias)))
        tk = ln(t.reshape(f0, T, C), net.norm_start)
        bi = 0
        for si, stage in enumerate(net.layers):
            res = tk
            for i, blk in enumerate(stage.blocks):
                xw, xs, xmap, a = block(blk, res, i, si, f"layers.{si}.blocks.{i}")
                att = t.cat_cols([t.reshape(xw, T, half), t.reshape(xs, T, half)])
                p = t.axpby(drop(ln(lin(att, a.proj), blk.norm1), 2 * bi), res)
                if net.local_connection:
                    cab = blk.conv.cab
                    c1 = t.unary(t.conv_im2col(xmap, cab[0].weight, cab[0].bias, N(cab[0].weight), N(cab[0].bias), 3), "gelu")
                    c2 = t.conv_im2col(c1, cab[2].weight, cab[2].bias, N(cab[2].weight), N(cab[2].bias), 3)
                    ca = cab[3].attention
                    xn = t.reshape(t.rcan_gate(c2, t.reshape(p, B, H, W, C), ca[1].weight, ca[1].bias, ca[3].weight, ca[3].bias,
                                               (N(ca[1].weight), N(ca[1].bias), N(ca[3].weight), N(ca[3].bias))), T, C)
                else:
                    xn = p
                m = lin(t.unary(lin(xn, blk.mlp.fc1), "gelu"), blk.mlp.fc2)
                res = t.axpby(drop(ln(m, blk.norm2), 2 * bi + 1), xn)
                bi += 1
            tk = t.reshape(t.conv(t.reshape(res, B, H, W, C), f"layers.{si}.conv", (N(stage.conv.weight), N(stage.conv.bias)),
                                  res=(t.reshape(tk, B, H, W, C), 1.0)), T, C)
        tk = ln(tk, net.norm_end)
        f = t.conv(t.reshape(tk, B, H, W, C), "conv_after_body", (N(net.conv_after_body.weight), N(net.conv_after_body.bias)),
                   res=(f0, 1.0))
        cb = net.conv_before_upsample[0]
        u = t.relu(t.conv(f, "conv_before_upsample.0", (N(cb.weight), N(cb.bias))), 0.01)
        for k, conv in enumerate(list(net.upsample.up)[0::2]):
            u = t.shuffle(t.conv(u, f"upsample.up.{2 * k}", (N(conv.weight), N(conv.bias))), 2)
        y = t.conv_out1(u, net.conv_last.weight, net.conv_last.bias, (N(net.conv_last.weight), N(net.conv_last.bias)))
        self.saved = (t, y)
        Bo, Ho, Wo = y.t.shape
        return y.t.view(Bo, 1, Ho, Wo)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        assert self.saved is not None, "backward() without a saved forward"
        assert not need_dx, "GRL: no gradient with respect to the input image"
        tape, out = self.saved
        tape.backward(out, dy.reshape(out.t.shape).contiguous(), grads)
        return None
