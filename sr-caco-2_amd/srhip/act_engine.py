"""ACT on libsrhip (reference dlib/models/network_act.py:321-541): a CNN branch (RCAN residual groups)
and a transformer branch (3 x 3 tokens, self-attention + cross-scale attention against overlapping 6 x 6 tokens) that
exchange features in four fusion blocks.  Written directly over the libsrhip ops: 3 x 3 convs on the split-MFMA conv
kernels, the 5 x 5 head convs as im2col (srhip_unfold) + GEMM, Linears on the split-MFMA GEMMs (weight planes: srhip/planes.py), 1 x 1 convs on the exact-f32 GEMM, the attention's
(sample, head) products as two batched launches (srhip_gemm_nt_batched) around srhip_softmax_rows, F.unfold / F.fold as srhip_unfold / srhip_fold, RCAN's channel
attention as srhip_channel_gate.  Training (save=True) runs the same graph on the tape of srhip/tape.py (_forward_tape): the
backward of every op is derived there from kernels of this library (the attention products transposed, F.fold / F.unfold as
each other's adjoints, srhip_layernorm_rows_bwd, srhip_softmax_rows_bwd)."""
import math

import torch

from . import ops
from .planes import PlaneCache
from .swinir_engine import _Bufs
from .tape import Tape, WeightBank


class ACTEngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.bank = WeightBank()
        self.planes = PlaneCache()
        self.bufs = _Bufs()
        self.prepared = False
        self.saved = None
        self.taps = None          # tests: dict that receives intermediate tensors (names as oracle.act_forward's taps)

    def invalidate(self):
        self.prepared = False
        self.planes.clear()

    def bucket_prefixes(self):
        return [[""]]

    # ------------------------------------------------------------------ weights
    def _convs3(self):
        net = self.net
        out = []
        for g in range(net.n_fusionblocks):
            rg = net.cnn_branch[g]
            for r in range(net.n_resblocks):
                out += [(f"cnn_branch.{g}.body.{r}.body.0", rg.body[r].body[0]), (f"cnn_branch.{g}.body.{r}.body.2", rg.body[r].body[2])]
            out.append((f"cnn_branch.{g}.body.{net.n_resblocks}", rg.body[net.n_resblocks]))
        for i in range(net.n_fusionblocks - 1):
            out += [(f"fusion_cnn.{i}.0", net.fusion_cnn[i][0]), (f"fusion_cnn.{i}.2", net.fusion_cnn[i][2])]
        out.append(("conv_last", net.conv_last))
        return out

    def prepare(self):
        net, bank = self.net, self.bank
        bank.begin()
        for key, m in self._convs3():
            bank.conv(key, m.weight, m.bias, "c3")
        F = net.n_feats
        for st in range(int(math.log2(net.upscale))):
            c = net.tail[0][2 * st]
            for j in range(4):
                bank.conv(f"tail.0.{2 * st}.{j}", c.weight[j * F:(j + 1) * F], c.bias[j * F:(j + 1) * F], "c3")
        bank.finish(next(net.parameters()).device)
        self.prepared = True

    # ------------------------------------------------------------------ pieces
    def _conv3(self, key, x, out=None, relu=False):
        e = self.bank.d[key]
        B, H, W, _ = x.shape
        y = torch.empty(B, H, W, e.Co, device=x.device) if out is None else out
        ops.conv3x3(x, e.wp, e.bias, e.Co, out=y)
        if relu:
            ops.leaky_relu_(y, 0.0)
        return y

    def _linear(self, x2, lin, out=None):
        """nn.Linear on the GEMM kernels, the weight as planes (srhip/planes.py: f32-grade; one product under --amp)"""
        w, b = self.planes.linear(id(lin), lin.weight.data, None if lin.bias is None else lin.bias.data)
        return ops.gemm_nt(x2, w, b, out=out)

    @staticmethod
    def _ln(x2, ln):
        return ops.layernorm_rows(x2, ln.weight.data, ln.bias.data, torch.empty_like(x2))

    @staticmethod
    def _gelu_(x):
        return ops.unary(x, x, "gelu")

    def _attend(self, q, k, v, B, Tq, Tk, heads, dh, scale):
        """q [B*Tq, heads*dh], k / v [B*Tk, heads*dh] (views with any row pitch) -> [B*Tq, heads*dh].  The (sample, head)
        products run as batches of one launch each (srhip_gemm_nt_batched), a chunk of samples at a time so that the
        attention matrices stay below 2 GiB."""
        out = torch.empty(B * Tq, heads * dh, device=q.device)
        Tk4 = (Tk + 3) & ~3                         # the GEMM takes contraction lengths / pitches that are multiples of 4
        nb = max(1, min(B, (1 << 29) // (heads * Tq * Tk4)))
        dots = torch.zeros(nb * heads, Tq, Tk4, device=q.device)      # pad columns stay 0 (softmax writes [:Tk] only)
        vt = torch.zeros(nb * heads, dh, Tk4, device=q.device)
        for b0 in range(0, B, nb):
            n = min(nb, B - b0)
            z = n * heads
            qs, ks, vs = q[b0 * Tq:(b0 + n) * Tq], k[b0 * Tk:(b0 + n) * Tk], v[b0 * Tk:(b0 + n) * Tk]
            vt[:z, :, :Tk].copy_(vs.reshape(n, Tk, heads, dh).permute(0, 2, 3, 1).reshape(z, dh, Tk))
            ops.gemm_nt_batched(qs[:, :dh], (Tq * qs.stride(0), dh), ks[:, :dh], (Tk * ks.stride(0), dh),
                                dots[0, :, :Tk], (heads * Tq * Tk4, Tq * Tk4), Tq, Tk, dh, z, heads)
            ops.softmax_rows_(dots[:z].view(z * Tq, Tk4)[:, :Tk], scale)
            o = out[b0 * Tq:(b0 + n) * Tq]
            ops.gemm_nt_batched(dots[0], (heads * Tq * Tk4, Tq * Tk4), vt[0], (heads * dh * Tk4, dh * Tk4),
                                o[:, :dh], (Tq * o.stride(0), dh), Tq, dh, Tk4, z, heads)
        return out

    def _self_attention(self, blk, x2, B, T):
        net = self.net
        qkv = self._linear(self._ln(x2, blk.norm), blk.fn.to_qkv)
        inner = net.n_heads * net.dim_head
        o = self._attend(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], B, T, T, net.n_heads, net.dim_head,
                         net.dim_head ** -0.5)
        return self._linear(o, blk.fn.to_out[0])

    def _cross_attention(self, blk, xq, xkv, B, Tq, Tk):
        net = self.net
        heads = net.n_heads // 2
        inner = heads * net.dim_head
        q = self._linear(self._ln(xq, blk.norm), blk.fn.to_q)
        kv = self._linear(self._ln(xkv, blk.norm2), blk.fn.to_kv)
        o = self._attend(q, kv[:, :inner], kv[:, inner:], B, Tq, Tk, heads, net.dim_head, net.dim_head ** -0.5)
        return self._linear(o, blk.fn.to_out[0])

    def _ffn(self, blk, x2):                       # PreNorm(FeedForward)
        h = self._gelu_(self._linear(self._ln(x2, blk.norm), blk.fn.net[0]))
        return self._linear(h, blk.fn.net[3])

    def _ln_mlp(self, seq, x2):                    # Sequential(LayerNorm, Linear, GELU, Linear)
        h = self._gelu_(self._linear(self._ln(x2, seq[0]), seq[1]))
        return self._linear(h, seq[3])

    def _conv5(self, conv, x, relu):
        """5 x 5 conv (padding 2) as im2col + GEMM"""
        B, H, W, C = x.shape
        Co = conv.weight.shape[0]
        w2 = conv.weight.data.reshape(Co, -1)
        y = torch.empty(B, H, W, Co, device=x.device)
        nb = max(1, min(B, (1 << 28) // (H * W * C * 25)))        # samples per im2col buffer (GEMM operands stay < 2 GiB)
        cols = torch.empty(nb * H * W, C * 25, device=x.device)
        for b0 in range(0, B, nb):
            b1 = min(B, b0 + nb)
            n = (b1 - b0) * H * W
            ops.unfold(x[b0:b1], C, 5, 1, 2, cols[:n])
            ops.gemm_nt(cols[:n], w2, conv.bias.data, out=y[b0:b1].view(n, Co))
        if relu:
            ops.leaky_relu_(y, 0.0)
        return y

    # ------------------------------------------------------------------ forward
    def forward(self, x3, dp=None, save=False):
        if not self.prepared:
            self.prepare()
        if save:
            return self._forward_tape(x3)
        net = self.net
        B, H, W = x3.shape
        dev = x3.device
        nf, ts, emb = net.n_feats, net.token_size, net.embedding_dim
        x = ops.conv3x3_cin1_fwd(x3, net.head[0].weight.data, net.head[0].bias.data, nf)
        for j in (1, 2):
            r = self._conv5(net.head[j].body[0], x, True)
            r = self._conv5(net.head[j].body[2], r, False)
            ops.axpby(r, x, 1.0, 1.0)
            x = r
        identity = x

        def tap(name, v, img=False):
            if self.taps is not None:
                self.taps[name] = (v.permute(0, 3, 1, 2) if img else v.view(B, -1, v.shape[-1])).detach().clone()
        tap("head", x, True)
        nTy, nTx = (H - ts) // ts + 1, (W - ts) // ts + 1
        T = nTy * nTx
        nLy, nLx = (H - 2 * ts) // ts + 1, (W - 2 * ts) // ts + 1
        TL = nLy * nLx
        tk = torch.empty(B * T, emb, device=dev)
        ops.unfold(x, nf, ts, ts, 0, tk)
        enc = self._linear(tk, net.linear_encoding)
        ops.axpby(tk, enc, 1.0, 1.0)
        tap("enc", tk)
        half = emb // 2
        ch = half // (ts * ts)
        f = None
        for i in range(net.n_fusionblocks):
            ops.axpby(tk, self._self_attention(net.mhsa_block[i][0], tk, B, T), 1.0, 1.0)
            tap(f"sa{i}", tk)
            ops.axpby(tk, self._ffn(net.mhsa_block[i][1], tk), 1.0, 1.0)
            tap(f"ffn{i}", tk)
            ta = tk[:, :half].contiguous()
            img = torch.empty(B, H, W, ch, device=dev)
            ops.fold(tk[:, half:], ch, ts, ts, img)
            tb = torch.empty(B * TL, ch * 4 * ts * ts, device=dev)
            ops.unfold(img, ch, 2 * ts, ts, 0, tb)
            cs = net.csta_block[i]
            tb = self._ln_mlp(cs[0], tb)
            ta_new = self._cross_attention(cs[1], ta, tb, B, T, TL)
            tb_new = self._cross_attention(cs[2], tb, ta, B, TL, T)
            ops.axpby(ta_new, ta, 1.0, 1.0)
            ops.axpby(tb_new, tb, 1.0, 1.0)
            tb = self._ln_mlp(cs[3], tb_new)
            ops.fold(tb, ch, 2 * ts, ts, img)
            tk = torch.empty(B * T, emb, device=dev)
            tk[:, :half].copy_(ta_new)
            ops.unfold(img, ch, ts, ts, 0, tk[:, half:])
            ops.axpby(tk, self._ffn(cs[4], tk), 1.0, 1.0)
            tap(f"csta{i}", tk)
            # CNN branch: ResidualGroup
            rg = net.cnn_branch[i]
            x0 = x
            for r in range(net.n_resblocks):
                m = rg.body[r]
                pre = f"cnn_branch.{i}.body.{r}.body"
                a = self._conv3(pre + ".0", x, relu=True)
                a = self._conv3(pre + ".2", a)
                ca = m.body[3].conv_du
                w1, w2 = ca[0].weight.data, ca[2].weight.data
                y = torch.empty_like(x)
                ops.channel_gate(a, w1.reshape(w1.shape[0], -1).contiguous(), ca[0].bias.data,
                                 w2.reshape(w2.shape[0], -1).contiguous(), ca[2].bias.data, x, a, y)
                x = y
            x = self._conv3(f"cnn_branch.{i}.body.{net.n_resblocks}", x)
            ops.axpby(x, x0, 1.0, 1.0)
            tap(f"cnn{i}", x, True)
            tk_res, x_res = tk, x
            f = torch.empty(B, H, W, 2 * nf, device=dev)
            f[..., :nf].copy_(x)
            ops.fold(tk, nf, ts, ts, f[..., nf:])
            f2 = f.view(B * H * W, 2 * nf)
            g = f2
            for j in range(4):
                fb = net.fusion_block[i][j]
                w0, w2 = fb.body[0].weight.data, fb.body[2].weight.data
                r = ops.gemm_nt(g, w0.reshape(w0.shape[0], -1).contiguous())
                ops.leaky_relu_(r, 0.0)
                r = ops.gemm_nt(r, w2.reshape(w2.shape[0], -1).contiguous())
                ops.axpby(r, g, 1.0, 1.0)
                g = r
            ops.axpby(f2, g, 1.0, 1.0)
            tap(f"f{i}", f, True)
            if i != net.n_fusionblocks - 1:
                # the reference splits f = cat(x, tokens) as "x_tkn, x = torch.split(f, n_feats, 1)" (:527): the FIRST half
                # (the CNN features) goes on as tokens, the second (the folded tokens) as the CNN branch's input
                tk = torch.empty(B * T, emb, device=dev)
                ops.unfold(f[..., :nf], nf, ts, ts, 0, tk)
                tk2 = self._ln_mlp(net.fusion_mlp[i], tk)
                ops.axpby(tk2, tk_res, 1.0, 1.0)
                tk = tk2
                xa = self._conv3(f"fusion_cnn.{i}.0", f[..., nf:], relu=True)
                xa = self._conv3(f"fusion_cnn.{i}.2", xa)
                ops.axpby(xa, x_res, 1.0, 1.0)
                x = xa
        x = self._conv3("conv_last", f)
        ops.axpby(x, identity, 1.0, 1.0)
        for st in range(int(math.log2(net.upscale))):
            u = torch.empty(B, x.shape[1], x.shape[2], 4 * nf, device=dev)
            for j in range(4):
                self._conv3(f"tail.0.{2 * st}.{j}", x, out=u[..., j * nf:(j + 1) * nf])
            x = ops.pixel_shuffle(u, 2, nhwc_out=True)
        y = ops.conv3x3_cout1_fwd(x, net.tail[1].weight.data, net.tail[1].bias.data)
        return y.view(B, 1, y.shape[1], y.shape[2])

    # ------------------------------------------------------------------ training: the same graph on the tape
    def _forward_tape(self, x3):
        """forward() op for op with the tape recording (srhip/tape.py): 3 x 3 convs on the bank's planes, everything on
        token rows (Linears, 1 x 1 convs, the attention products) on the exact-f32 GEMMs."""
        net = self.net
        t = Tape(self.bufs, self.bank, True, x3.device)
        nm = {id(p): k for k, p in net.named_parameters()}
        N = lambda p: nm[id(p)]
        B, H, W = x3.shape
        nf, ts, emb = net.n_feats, net.token_size, net.embedding_dim
        half = emb // 2
        ch = half // (ts * ts)
        nTy, nTx = (H - ts) // ts + 1, (W - ts) // ts + 1
        T = nTy * nTx
        TL = ((H - 2 * ts) // ts + 1) * ((W - 2 * ts) // ts + 1)

        def lin(x, m):
            return t.linear(x, m.weight, m.bias, N(m.weight), None if m.bias is None else N(m.bias))

        def ln(x, m):
            return t.layernorm_rows(x, m, N(m.weight), N(m.bias))

        def c3(x, key, mod, **kw):
            return t.conv(x, key, (N(mod.weight), N(mod.bias)), **kw)

        def self_attention(blk, x):
            qkv = lin(ln(x, blk.norm), blk.fn.to_qkv)
            inner = net.n_heads * net.dim_head
            o = t.attend(t.cols(qkv, 0, inner), t.cols(qkv, inner, 2 * inner), t.cols(qkv, 2 * inner, 3 * inner), B, T, T,
                         net.n_heads, net.dim_head, net.dim_head ** -0.5)
            return lin(o, blk.fn.to_out[0])

        def cross_attention(blk, xq, xkv, Tq, Tk):
            heads = net.n_heads // 2
            inner = heads * net.dim_head
            q = lin(ln(xq, blk.norm), blk.fn.to_q)
            kv = lin(ln(xkv, blk.norm2), blk.fn.to_kv)
            o = t.attend(q, t.cols(kv, 0, inner), t.cols(kv, inner, 2 * inner), B, Tq, Tk, heads, net.dim_head, net.dim_head ** -0.5)
            return lin(o, blk.fn.to_out[0])

        def ffn(blk, x):
            return lin(t.unary(lin(ln(x, blk.norm), blk.fn.net[0]), "gelu"), blk.fn.net[3])

        def ln_mlp(seq, x):
            return lin(t.unary(lin(ln(x, seq[0]), seq[1]), "gelu"), seq[3])

        xv = t.conv_in1(x3, net.head[0].weight, net.head[0].bias, (N(net.head[0].weight), N(net.head[0].bias)))
        for j in (1, 2):
            b0, b2 = net.head[j].body[0], net.head[j].body[2]
            r = t.conv_im2col(xv, b0.weight, b0.bias, N(b0.weight), N(b0.bias), 5, relu=True)
            r = t.conv_im2col(r, b2.weight, b2.bias, N(b2.weight), N(b2.bias), 5)
            xv = t.axpby(r, xv)
        identity = xv
        tk = t.unfold(xv, ts, ts)
        tk = t.axpby(tk, lin(tk, net.linear_encoding))
        f = None
        for i in range(net.n_fusionblocks):
            tk = t.axpby(tk, self_attention(net.mhsa_block[i][0], tk))
            tk = t.axpby(tk, ffn(net.mhsa_block[i][1], tk))
            ta = t.cols(tk, 0, half)
            img = t.fold(t.cols(tk, half, emb), ch, ts, ts, B, H, W)
            tb = t.unfold(img, 2 * ts, ts)
            cs = net.csta_block[i]
            tb = ln_mlp(cs[0], tb)
            ta_new = t.axpby(cross_attention(cs[1], ta, tb, T, TL), ta)
            tb_new = t.axpby(cross_attention(cs[2], tb, ta, TL, T), tb)
            tb = ln_mlp(cs[3], tb_new)
            img = t.fold(tb, ch, 2 * ts, ts, B, H, W)
            tk = t.cat_cols([ta_new, t.unfold(img, ts, ts)])
            tk = t.axpby(tk, ffn(cs[4], tk))
            rg = net.cnn_branch[i]
            x0 = xv
            for r in range(net.n_resblocks):
                m = rg.body[r]
                pre = f"cnn_branch.{i}.body.{r}.body"
                a = c3(xv, pre + ".0", m.body[0], relu=True)
                a = c3(a, pre + ".2", m.body[2])
                ca = m.body[3].conv_du
                xv = t.rcan_gate(a, xv, ca[0].weight, ca[0].bias, ca[2].weight, ca[2].bias,
                                 (N(ca[0].weight), N(ca[0].bias), N(ca[2].weight), N(ca[2].bias)))
            last = rg.body[net.n_resblocks]
            xv = c3(xv, f"cnn_branch.{i}.body.{net.n_resblocks}", last, res=(x0, 1.0))
            tk_res, x_res = tk, xv
            f = t.cat_cols([xv, t.fold(tk, nf, ts, ts, B, H, W)])
            f2 = t.reshape(f, B * H * W, 2 * nf)
            g = f2
            for j in range(4):
                fb = net.fusion_block[i][j]
                r = t.relu(t.linear(g, fb.body[0].weight, None, N(fb.body[0].weight)))
                g = t.axpby(t.linear(r, fb.body[2].weight, None, N(fb.body[2].weight)), g)
            f2 = t.axpby(f2, g)
            f = t.reshape(f2, B, H, W, 2 * nf)
            if i != net.n_fusionblocks - 1:
                # "x_tkn, x = torch.split(f, n_feats, 1)" (:527): the CNN half goes on as tokens, the token half as the map
                tk = t.unfold(t.cols(f, 0, nf), ts, ts)
                tk = t.axpby(ln_mlp(net.fusion_mlp[i], tk), tk_res)
                fc = net.fusion_cnn[i]
                xa = c3(t.cols(f, nf, 2 * nf), f"fusion_cnn.{i}.0", fc[0], relu=True)
                xv = c3(xa, f"fusion_cnn.{i}.2", fc[2], res=(x_res, 1.0))
        xv = c3(f, "conv_last", net.conv_last, res=(identity, 1.0))
        F = nf
        for st in range(int(math.log2(net.upscale))):
            c = net.tail[0][2 * st]
            parts = [t.conv(xv, f"tail.0.{2 * st}.{j}", (N(c.weight), N(c.bias), (j * F, (j + 1) * F))) for j in range(4)]
            xv = t.shuffle(t.cat_cols(parts), 2)
        out = t.conv_out1(xv, net.tail[1].weight, net.tail[1].bias, (N(net.tail[1].weight), N(net.tail[1].bias)))
        self.saved = (t, out)
        Bo, Ho, Wo = out.t.shape
        return out.t.view(Bo, 1, Ho, Wo)

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        assert self.saved is not None, "backward() without a saved forward"
        assert not need_dx, "ACT: no gradient with respect to the input image"
        tape, out = self.saved
        tape.backward(out, dy.reshape(out.t.shape).contiguous(), grads)
        return None
