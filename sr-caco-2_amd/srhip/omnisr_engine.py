"""OmniSR on libsrhip (reference dlib/models/network_omni_sr.py:430-591): OSAG groups of omni
self-attention blocks -- MBConv with squeeze-excitation, 8x8-window and grid attention with relative-position bias, the
channel attention of a window / of a grid position, gated depthwise feed-forwards -- and ESA behind every group.  Written
directly over the libsrhip ops: 1x1 convs and Linears on the exact-f32 GEMM, 3x3 convs on the conv kernels (ESA's stride-2
conv as srhip_unfold + GEMM), the rest in omni_ops.hip.  Window / grid token orders are produced by permuted copies (data
movement only).  Training (save=True) runs the same graph on the tape of srhip/tape.py (_forward_tape), its backward composed
from kernels of this library."""
import torch

from . import ops
from .swinir_engine import _Bufs
from .tape import Tape, WeightBank


def _w2(conv):
    w = conv.weight.data
    return w.reshape(w.shape[0], -1).contiguous()


def _b(m):
    return None if m.bias is None else m.bias.data


class OmniSREngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.prepared = True
        self.saved = None
        self.taps = None
        self._bias = {}
        self.bufs = _Bufs()
        self.bank = WeightBank()
        self._bank_ready = False

    def invalidate(self):
        self._bias = {}
        self._bank_ready = False

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
            return self._forward_tape(x3)
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

    # ------------------------------------------------------------------ training: the same graph on the tape
    def _forward_tape(self, x3):
        """forward() op for op with the tape recording (srhip/tape.py).  The window / grid attention runs as batched GEMMs
        around the row softmax (Tape.attend, the relative-position bias as a periodic addend), the channel attention likewise on
        L2-normalised rows; depthwise convs, the gates, pooling and the resize carry the backward ops written next to them."""
        net = self.net
        B, H, W = x3.shape
        ws, s, nf = net.window_size, net.upscale, net.num_feat
        if H % ws or W % ws:
            raise NotImplementedError("OmniSR on libsrhip: training patches have to be multiples of the 8-pixel window")
        X, Y = H // ws, W // ws
        P = B * H * W
        if not self._bank_ready:
            self.bank.begin()
            self.bank.conv("output", net.output.weight, net.output.bias, "c3")
            self.bank.conv("up.0", net.up[0].weight, net.up[0].bias, "c3")
            for g, osag in enumerate(net.residual_layer):
                self.bank.conv(f"esa{g}.conv3", osag.esa.conv3.weight, osag.esa.conv3.bias, "c3")
            self.bank.finish(x3.device)
            self._bank_ready = True
        t = Tape(self.bufs, self.bank, True, x3.device)
        nm = {id(p): k for k, p in net.named_parameters()}
        N = lambda p: None if p is None else nm[id(p)]

        def conv1(x, m):                                  # 1 x 1 conv on an NHWC map
            Bc, Hc, Wc, C = x.t.shape
            y = t.linear(t.reshape(x, Bc * Hc * Wc, C), m.weight, m.bias, N(m.weight), N(m.bias))
            return t.reshape(y, Bc, Hc, Wc, m.weight.shape[0])

        def ln2d(x, m):
            Bc, Hc, Wc, C = x.t.shape
            return t.reshape(t.layernorm_rows(t.reshape(x, Bc * Hc * Wc, C), m, N(m.weight), N(m.bias), eps=1e-6), Bc, Hc, Wc, C)

        def mbconv(m, x):
            fn = m.fn
            h = t.unary(conv1(x, fn[0]), "gelu")
            d = t.unary(t.dwconv(h, fn[2].weight, fn[2].bias, N(fn[2].weight), N(fn[2].bias)), "gelu")
            g = t.se_gate(d, fn[4].gate[1].weight, fn[4].gate[3].weight, (N(fn[4].gate[1].weight), N(fn[4].gate[3].weight)))
            return t.axpby(conv1(g, fn[5]), x)

        def attention(m, x, grid):
            C = x.t.shape[3]
            if not grid:    # 'b d (x w1) (y w2) -> b x y w1 w2 d'
                fwd = lambda v: v.reshape(B, X, ws, Y, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * X * Y * ws * ws, C)
                inv = lambda v: v.reshape(B, X, Y, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
            else:           # 'b d (w1 x) (w2 y) -> b x y w1 w2 d'
                fwd = lambda v: v.reshape(B, ws, X, ws, Y, C).permute(0, 2, 4, 1, 3, 5).reshape(B * X * Y * ws * ws, C)
                inv = lambda v: v.reshape(B, X, Y, ws, ws, C).permute(0, 3, 1, 4, 2, 5).reshape(B, H, W, C)
            tk = t.relayout(x, fwd, inv)
            qkv = t.linear(t.layernorm_rows(tk, m.norm, N(m.norm.weight), N(m.norm.bias)), m.fn.to_qkv.weight, None,
                           N(m.fn.to_qkv.weight))
            heads = m.fn.heads
            bias = on_dbias = None
            if m.fn.with_pe:
                tab = m.fn.rel_pos_bias.weight
                idx = m.fn.rel_pos_indices.to(tab.device)
                bias = tab.data[idx].permute(2, 0, 1).contiguous()                    # [heads, 64, 64]

                def on_dbias(d, tab=tab, idx=idx):
                    def prod(o):
                        o.zero_()
                        o.index_add_(0, idx.reshape(-1), d.permute(1, 2, 0).reshape(-1, heads))
                    t.gparam(N(tab), prod)
            o = t.attend(t.cols(qkv, 0, C), t.cols(qkv, C, 2 * C), t.cols(qkv, 2 * C, 3 * C), B * X * Y, ws * ws, ws * ws, heads,
                         C // heads, (C // heads) ** -0.5, bias=bias, on_dbias=on_dbias)
            y = t.axpby(t.linear(o, m.fn.to_out[0].weight, None, N(m.fn.to_out[0].weight)), tk)
            return t.relayout(y, inv, fwd)

        def ffn(m, x):
            p = conv1(ln2d(x, m.norm), m.fn.project_in)
            d = t.dwconv(p, m.fn.dwconv.weight, m.fn.dwconv.bias, N(m.fn.dwconv.weight), N(m.fn.dwconv.bias))
            return t.axpby(conv1(t.gelu_gate(d), m.fn.project_out), x)

        def channel_attention(m, x, grid):
            C = x.t.shape[3]
            heads, ps = m.fn.heads, m.fn.ps
            dch = C // heads
            hh, ww = H // ps, W // ps
            q0 = conv1(ln2d(x, m.norm), m.fn.qkv)
            d = t.dwconv(q0, m.fn.qkv_dwconv.weight, m.fn.qkv_dwconv.bias, N(m.fn.qkv_dwconv.weight), N(m.fn.qkv_dwconv.bias))
            if not grid:    # 'b (head d) (h ph) (w pw) -> b (h w) head d (ph pw)'
                n, G = ps * ps, B * hh * ww
                fwd = lambda v: v.reshape(B, hh, ps, ww, ps, heads, dch).permute(0, 1, 3, 5, 6, 2, 4).reshape(G * heads * dch, n)
                inv = lambda v: v.reshape(B, hh, ww, heads, dch, ps, ps).permute(0, 1, 5, 2, 6, 3, 4).reshape(B, H, W, C)
            else:           # 'b (head d) (h ph) (w pw) -> b (ph pw) head d (h w)'
                n, G = hh * ww, B * ps * ps
                fwd = lambda v: v.reshape(B, hh, ps, ww, ps, heads, dch).permute(0, 2, 4, 5, 6, 1, 3).reshape(G * heads * dch, n)
                inv = lambda v: v.reshape(B, ps, ps, heads, dch, hh, ww).permute(0, 5, 1, 6, 2, 3, 4).reshape(B, H, W, C)
            n4 = (n + 3) & ~3
            q, k, v = (t.relayout(t.cols(d, i * C, (i + 1) * C), fwd, inv, pad_last=n4) for i in range(3))
            temp = m.fn.temperature
            s_rows = temp.data.reshape(1, heads, 1).expand(G, heads, dch).reshape(-1).contiguous()

            def on_temp(ds):
                t.gparam(N(temp), lambda o: o.view(heads).copy_(ds.view(G, heads, dch).sum((0, 2))))
            qs = t.scale_rows(t.normalize_rows(q), s_rows, on_temp)
            o = t.attend(qs, t.normalize_rows(k), v, G * heads, dch, dch, 1, n4, 1.0)
            back = t.relayout(o, lambda u: inv(u[..., :n]), lambda u: torch.nn.functional.pad(fwd(u), (0, n4 - n)))
            return t.axpby(conv1(back, m.fn.project_out), x)

        def esa(g, m, x):
            c1_ = conv1(x, m.conv1)
            c1 = t.conv_patches(c1_, m.conv2.weight, m.conv2.bias, N(m.conv2.weight), N(m.conv2.bias), 3, 2)
            c3 = t.conv(t.maxpool(c1, 7, 3), f"esa{g}.conv3", (N(m.conv3.weight), N(m.conv3.bias)))
            c3 = t.bilinear(c3, H, W)
            c4 = conv1(t.axpby(c3, conv1(c1_, m.conv_f)), m.conv4)
            return t.mul_sigmoid(x, c4)

        residual = t.conv_in1(x3, net.input.weight, net.input.bias, (N(net.input.weight), N(net.input.bias)))
        out = residual
        for g, osag in enumerate(net.residual_layer):
            gin = out
            nblk = len(osag.residual_layer) - 1
            for bk in range(nblk):
                L = osag.residual_layer[bk].layer
                out = mbconv(L[0], out)
                out = attention(L[2], out, False)
                out = ffn(L[4], out)
                out = channel_attention(L[5], out, False)
                out = ffn(L[6], out)
                out = attention(L[8], out, True)
                out = ffn(L[10], out)
                out = channel_attention(L[11], out, True)
                out = ffn(L[12], out)
            out = t.axpby(conv1(out, osag.residual_layer[nblk]), gin)
            out = esa(g, osag.esa, out)
        out = t.conv(out, "output", (N(net.output.weight), N(net.output.bias)), res=(residual, 1.0))
        u = t.conv(out, "up.0", (N(net.up[0].weight), N(net.up[0].bias)))
        y = t.reshape(t.shuffle(u, s), B, H * s, W * s)
        self.saved = (t, y)
        return y.t.view(B, 1, H * s, W * s)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        assert self.saved is not None, "backward() without a saved forward"
        assert not need_dx, "OmniSR: no gradient with respect to the input image"
        tape, out = self.saved
        tape.backward(out, dy.reshape(out.t.shape).contiguous(), grads)
        return None
