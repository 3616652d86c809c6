"""OmniSR on libsrhip, evaluation forward (reference dlib/models/network_omni_sr.py:430-591): OSAG groups of omni
self-attention blocks -- MBConv with squeeze-excitation, 8x8-window and grid attention with relative-position bias, the
channel attention of a window / of a grid position, gated depthwise feed-forwards -- and ESA behind every group.  Written
directly over the libsrhip ops: 1x1 convs and Linears on the exact-f32 GEMM, 3x3 convs on the conv kernels (ESA's stride-2
conv as srhip_unfold + GEMM), the rest in omni_ops.hip.  Window / grid token orders are produced by permuted copies (data
movement only).  Inference only."""
import torch

from . import ops


def _w2(conv):
    w = conv.weight.data
    return w.reshape(w.shape[0], -1).contiguous()


def _b(m):
    return None if m.bias is None else m.bias.data


class OmniSREngine:
    def __init__(self, net):
        self.net = net
        self.prepared = True
        self.saved = None
        self.taps = None
        self._bias = {}

    def invalidate(self):
        self._bias = {}

    def bucket_prefixes(self):
        return [[""]]

    # ------------------------------------------------------------------ pieces (x NHWC [B, H, W, C])
    @staticmethod
    def _conv1(x, conv):
        B, H, W, C = x.shape
        y = ops.gemm_nt(x.reshape(B * H * W, C), _w2(conv), _b(conv))
        return y.view(B, H, W, -1)

    @staticmethod
    def _conv3(x, conv):
        B, H, W, C = x.shape
        wp = torch.empty(9, conv.weight.shape[0], C, device=x.device)
        ops.pack_conv_weight(conv.weight.data, wp, None)
        return ops.conv3x3(x, wp, _b(conv), conv.weight.shape[0])

    @staticmethod
    def _ln2d(x, norm):
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        ops.layernorm_rows(x.view(-1, C), norm.weight.data, norm.bias.data, y.view(-1, C), eps=1e-6)
        return y

    def _mbconv(self, m, x):
        fn = m.fn
        B, H, W, C = x.shape
        h = self._conv1(x, fn[0])
        ops.unary(h, h, "gelu")
        d = torch.empty_like(h)
        ops.dwconv3x3(h, fn[2].weight.data, _b(fn[2]), d)
        ops.unary(d, d, "gelu")
        g = torch.empty_like(d)
        ops.channel_gate(d, fn[4].gate[1].weight.data, None, fn[4].gate[3].weight.data, None, None, d, g, mid_act="silu")
        y = self._conv1(g, fn[5])
        ops.axpby(y, x, 1.0, 1.0)
        return y

    def _attention(self, m, x, grid):
        """PreNormResidual(Attention) on the window (grid = False) or grid token order"""
        net = self.net
        B, H, W, C = x.shape
        ws = net.window_size
        X, Y = H // ws, W // ws
        if not grid:        # 'b d (x w1) (y w2) -> b x y w1 w2 d'
            t = x.view(B, X, ws, Y, ws, C).permute(0, 1, 3, 2, 4, 5)
        else:               # 'b d (w1 x) (w2 y) -> b x y w1 w2 d'
            t = x.view(B, ws, X, ws, Y, C).permute(0, 2, 4, 1, 3, 5)
        t = t.reshape(B * X * Y * ws * ws, C)
        tn = ops.layernorm_rows(t, m.norm.weight.data, m.norm.bias.data, torch.empty_like(t))
        qkv = ops.gemm_nt(tn, m.fn.to_qkv.weight.data)
        heads = m.fn.heads
        bias = None
        if m.fn.with_pe:
            key = id(m)
            if key not in self._bias:
                tab = m.fn.rel_pos_bias.weight.data                       # [(2 ws - 1)^2, heads]
                self._bias[key] = tab[m.fn.rel_pos_indices.to(tab.device)].permute(2, 0, 1).contiguous()
            bias = self._bias[key]
        o = torch.empty_like(t)
        ops.group_attention(qkv, bias, o, ws * ws, heads, (C // heads) ** -0.5)
        y = ops.gemm_nt(o, m.fn.to_out[0].weight.data)
        ops.axpby(y, t, 1.0, 1.0)
        y = y.view(B, X, Y, ws, ws, C)
        if not grid:
            return y.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        return y.permute(0, 3, 1, 4, 2, 5).reshape(B, H, W, C)

    def _ffn(self, m, x):
        B, H, W, C = x.shape
        p = self._conv1(self._ln2d(x, m.norm), m.fn.project_in)
        d = torch.empty_like(p)
        ops.dwconv3x3(p, m.fn.dwconv.weight.data, _b(m.fn.dwconv), d)
        g = torch.empty(B * H * W, p.shape[3] // 2, device=x.device)
        ops.gelu_gate(d.view(B * H * W, -1), g)
        y = self._conv1(g.view(B, H, W, -1), m.fn.project_out)
        ops.axpby(y, x, 1.0, 1.0)
        return y

    def _channel_attention(self, m, x, grid):
        B, H, W, C = x.shape
        q = self._conv1(self._ln2d(x, m.norm), m.fn.qkv)
        d = torch.empty_like(q)
        ops.dwconv3x3(q, m.fn.qkv_dwconv.weight.data, _b(m.fn.qkv_dwconv), d)
        o = torch.empty_like(x)
        ops.channel_attention(d, m.fn.temperature.data.reshape(-1).contiguous(), o, m.fn.heads, m.fn.ps, grid)
        y = self._conv1(o, m.fn.project_out)
        ops.axpby(y, x, 1.0, 1.0)
        return y

    def _esa(self, m, x):
        B, H, W, C = x.shape
        c1_ = self._conv1(x, m.conv1)
        f = c1_.shape[3]
        Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        cols = torch.empty(B * Ho * Wo, f * 9, device=x.device)
        ops.unfold(c1_, f, 3, 2, 0, cols)
        c1 = ops.gemm_nt(cols, _w2(m.conv2), _b(m.conv2)).view(B, Ho, Wo, f)
        vmax = ops.maxpool2d(c1, 7, 3)
        c3 = self._conv3(vmax, m.conv3)
        c3 = ops.bilinear_resize(c3, H, W)
        cf = self._conv1(c1_, m.conv_f)
        ops.axpby(c3, cf, 1.0, 1.0)
        c4 = self._conv1(c3, m.conv4)
        return ops.mul_sigmoid(x, c4, torch.empty_like(x))

    # ------------------------------------------------------------------ forward
    def forward(self, x3, dp=None, save=False):
        if save:
            raise NotImplementedError("OmniSR on libsrhip: inference only (BASELINE config 5's evaluation sweep); no backward")
        net = self.net
        B, H0, W0 = x3.shape
        ws, s = net.window_size, net.upscale
        ph, pw = (ws - H0 % ws) % ws, (ws - W0 % ws) % ws
        if ph or pw:
            x3 = torch.nn.functional.pad(x3, (0, pw, 0, ph))               # zero padding (check_image_size :568-575)
        x3 = x3.contiguous()
        B, H, W = x3.shape
        nf = net.num_feat

        def tap(name, v):
            if self.taps is not None:
                self.taps[name] = v.permute(0, 3, 1, 2).detach().clone()
        residual = ops.conv3x3_cin1_fwd(x3, net.input.weight.data, _b(net.input), nf)
        tap("input", residual)
        out = residual
        for g, osag in enumerate(net.residual_layer):
            gin = out
            nblk = len(osag.residual_layer) - 1
            for bk in range(nblk):
                L = osag.residual_layer[bk].layer
                out = self._mbconv(L[0], out); tap(f"g{g}b{bk}.mb", out)
                out = self._attention(L[2], out, False); tap(f"g{g}b{bk}.att1", out)
                out = self._ffn(L[4], out); tap(f"g{g}b{bk}.ffn1", out)
                out = self._channel_attention(L[5], out, False); tap(f"g{g}b{bk}.ca1", out)
                out = self._ffn(L[6], out)
                out = self._attention(L[8], out, True); tap(f"g{g}b{bk}.att2", out)
                out = self._ffn(L[10], out)
                out = self._channel_attention(L[11], out, True); tap(f"g{g}b{bk}.ca2", out)
                out = self._ffn(L[12], out); tap(f"g{g}b{bk}.out", out)
            out = self._conv1(out, osag.residual_layer[nblk])
            ops.axpby(out, gin, 1.0, 1.0)
            out = self._esa(osag.esa, out); tap(f"g{g}.esa", out)
        out = self._conv3(out, net.output)
        ops.axpby(out, residual, 1.0, 1.0)
        u = self._conv3(out, net.up[0])                                      # [B, H, W, in_chans * s * s]
        y = ops.pixel_shuffle(u, s)                                          # NCHW [B, 1, H s, W s]
        return y[:, :, :H0 * s, :W0 * s].contiguous()

    def backward(self, *a, **k):
        raise NotImplementedError("OmniSR on libsrhip: inference only (BASELINE config 5's evaluation sweep); no backward")
