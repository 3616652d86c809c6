"""GRL on libsrhip, evaluation forward (reference dlib/models/network_grl.py:1462-1512): stages of mixed-attention blocks
(:1061-1076) -- one qkv Linear whose halves feed an 8x8 (shifted) window attention and an anchored stripe attention
(average-pooled 4x4 anchors attend to the 8x8 stripe, the stripe attends back to the anchors), both cosine attentions with a
clamped logit scale and a 16 sigmoid(CPB MLP) bias, a shared output projection, post-norm residuals, a conv + GELU + conv +
channel-attention local branch, a GELU MLP -- each stage closed by a 3x3 conv and a skip, and a pixel-shuffle tail.  Written
directly over the libsrhip ops: Linears and 3x3 convs on the GEMM / conv kernels (fp16x2 / bf16x3 planes from 64 channels on
-- the kernels --amp narrows to one product -- exact f32 below; GELU as the fc2 prologue; the
C/4 channels of the local branch zero-padded to a multiple of 4: exact), attentions / pooling / bias images in grl_ops.hip,
the channel gate and LayerNorms of the earlier nets.  Tokens stay channels-last [B, H, W, C] throughout, so blc<->bchw,
roll, window_partition and window_reverse are address arithmetic inside the kernels.  Inference only."""
import torch
import torch.nn.functional as F

from . import ops
from .planes import PlaneCache


def _pad4(n):
    return (n + 3) // 4 * 4


class GRLEngine:
    def __init__(self, net):
        self.net = net
        self.prepared = True
        self.saved = None
        self.taps = None
        self._w = {}
        self.planes = PlaneCache()

    def invalidate(self):
        self._w = {}
        self.planes.clear()

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
            raise NotImplementedError("GRL on libsrhip: inference only (BASELINE config 5's evaluation sweep); no backward")
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

    def backward(self, *a, **k):
        raise NotImplementedError("GRL on libsrhip: inference only (BASELINE config 5's evaluation sweep); no backward")
