"""SwinIR forward / backward as a fixed sequence of libsrhip launches.

Mirrors SwinIR.forward (reference dlib/models/network_swinir.py:930-970) and the
autograd graph PyTorch would build for it, for the 'pixelshuffledirect'
upsampler on 1-channel inputs.  Everything stays in token order
[B*H*W][C] (= NHWC), so PatchEmbed / PatchUnEmbed transposes (:610-614,
:651-655), torch.roll and window_partition / window_reverse (:297-331) never
materialise.  LayerNorm affine parameters in front of a Linear are folded into
that Linear's weights once per step (W_f = W*gamma, b_f = b + W.beta); the
normalisation itself is the A-operand prologue of the GEMM, GELU the prologue
of fc2, residual adds and DropPath scaling the GEMM epilogues.

Per Swin block: forward = 2 stats kernels + 4 GEMMs + 1 attention kernel;
backward = 4 data-gradient GEMMs + 4 weight-gradient GEMMs (+ slice reducers)
+ 2 LayerNorm-backward kernels + 1 attention kernel.
"""
import os

import torch

from . import ops


class _Bufs:
    """Persistent activation / gradient buffers keyed by name (no allocation in
    the steady-state step; also what makes the step graph-capturable)."""

    def __init__(self):
        self.d = {}

    def get(self, key, *shape, device="cuda", dtype=torch.float32):
        t = self.d.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            if t is not None:
                ops.note_realloc()      # a replaced buffer is freed: captured graphs holding its address must be re-made
            t = torch.empty(shape, device=device, dtype=dtype)
            self.d[key] = t
        return t

    def clear(self):
        if self.d:
            ops.note_realloc()
        self.d.clear()


def net_dev(net):
    return net.conv_first.weight.device


class SwinIREngine:
    train_graph_default = True      # ModelPlain replays the training step from a hipGraph (TrainStep.step_graph)

    def __init__(self, net):
        self.net = net
        self.C = net.embed_dim
        self.hid = int(net.embed_dim * net.mlp_ratio)
        self.scale = net.upscale
        self.ci = net.in_chans          # 1: the single-channel edge kernels; otherwise _prepare_edges
        self.blocks = list(net.swin_blocks())
        self.direct = net.upsampler == "pixelshuffledirect"
        self.nearest = net.upsampler == "nearest_conv"      # 2 x [nearest x2, conv, LeakyReLU], conv_hr, conv_last (x4)
        self.stages = 0 if (self.direct or self.nearest) else int(round(__import__("math").log2(net.upscale)))
        # resi_connection '3conv' (network_swinir.py:545-552): conv C -> C/4, LeakyReLU(0.2), conv1x1, LeakyReLU(0.2),
        # conv C/4 -> C.  It runs on the SAME kernels as '1conv' with the C/4 channels zero-padded to a width they take
        # (45 -> 64 on the bf16x3 kernels, a multiple of 4 on the exact-f32 ones): padded weights / biases are 0,
        # LeakyReLU(0) = 0, so the padded channels carry exact zeros forward and backward.
        self.conv3 = getattr(net, "resi_connection", "1conv") == "3conv"
        self.c4 = self.C // 4
        self.c4p = (max(64, (self.c4 + 3) // 4 * 4) if ops.bx3_nt_for(self.C) else (self.c4 + 3) // 4 * 4)
        self.layer_of_block = []
        for li, layer in enumerate(net.layers):
            for _ in layer.residual_group.blocks:
                self.layer_of_block.append(li)
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()      # matmul operands: f32 tensors or bf16x3 planes
        self._prep = self._prep_sig = None
        self.prepared = False
        self.saved = None
        # The MLP half of a block as one kernel per direction on the Linear GEMMs' own two-plane fp16 operands (mlp_f16.hip; no extra weight planes): one
        # A fetch and one epilogue per direction instead of two, the hidden activation never read back.  SRHIP_MLP_F16=0:
        # the separate Linear launches.
        self.fuse_mlp_h = ops.mlp_f16_fusable(self.C, self.hid) and os.environ.get("SRHIP_MLP_F16", "1") != "0"
        # The W-MSA half, forward, as one kernel per block (wmsa_f16.hip): qkv Linear + window attention + proj Linear +
        # residual; q, k, v and the attention output are read back from L2 by the block that wrote them.  SRHIP_WMSA_F16=0:
        # the three separate launches.
        # The proj Linear's data gradient chained behind the fused MLP backward (SRHIP_CHAIN_PROJ=0: its own launch)
        self.chain_proj = os.environ.get("SRHIP_CHAIN_PROJ", "1") != "0"
        # ... and the qkv Linear's data gradient + LayerNorm backward of the block behind in FRONT of it (SRHIP_FRONT_QKV=0)
        self.front_qkv = os.environ.get("SRHIP_FRONT_QKV", "1") != "0"
        self.fuse_wmsa = (all(ops.wmsa_f16_fusable(self.C, b.num_heads) for b in self.blocks)
                          and os.environ.get("SRHIP_WMSA_F16", "1") != "0")

    def _qkv_bias(self, blk):
        """the qkv Linear's bias, or zeros when the net was built with qkv_bias=False (network_swinir.py:104)"""
        if blk.attn.qkv.bias is not None:
            return blk.attn.qkv.bias.data
        z = self.derived.get("zero.bq", 3 * self.C, device=blk.attn.qkv.weight.device)
        z.zero_()
        return z

    def bucket_prefixes(self):
        """Gradient buckets in backward-completion order: one per RSTB layer (the
        tail -- norm / conv_after_body / upsample -- rides with the last layer), the
        head (conv_first / patch_embed) last.  backward() calls on_layer_done(i)
        when bucket i is complete, for every bucket but the last."""
        n = len(self.net.layers)
        out = []
        for li in reversed(range(n)):
            pf = [f"layers.{li}."]
            if li == n - 1:
                pf += ["norm.", "conv_after_body.", "conv_before_upsample.", "upsample.", "conv_up1.", "conv_up2.",
                       "conv_hr.", "conv_last."]
            out.append(pf)
        out.append(["conv_first.", "patch_embed."])
        groups = self.layer0_bucket_groups()
        if groups is not None and n > 1:
            # Layer 0 is the LAST layer of backward: as one bucket (7.5 MB of the README net) it is announced 0.2 ms before
            # the step ends and its all-reduce is the one exchange a multi-GPU run cannot hide.  Split: the
            # upper blocks (+ the layer's conv), the middle blocks, and block 0 with the head -- each a contiguous range
            # of the flat buffer (named_parameters order: conv_first, patch_embed, layers.0.residual_group.blocks.0..,
            # layers.0.conv) announced as soon as ITS weight gradients are enqueued; the last bucket is 1.3 MB.
            first, middle, last = groups
            out = out[:n - 1]
            out.append([f"layers.0.residual_group.blocks.{j}." for j in first] + ["layers.0.conv."])
            if middle:
                out.append([f"layers.0.residual_group.blocks.{j}." for j in middle])
            out.append(["conv_first.", "patch_embed.", "absolute_pos_embed"]
                       + [f"layers.0.residual_group.blocks.{j}." for j in last])
        return out

    def layer0_bucket_groups(self):
        """(first, middle, last) block indices of layer 0's gradient buckets under data parallelism, or None (one bucket).
        SRHIP_DDP_SPLIT0=0 keeps the layer whole."""
        nb = len(self.net.layers[0].residual_group.blocks)
        if nb < 2 or os.environ.get("SRHIP_DDP_SPLIT0", "1") == "0":
            return None
        lo = max(1, nb // 2)
        return list(range(lo, nb)), list(range(1, lo)), [0]

    # ------------------------------------------------------------------ weights
    def invalidate(self):
        self.prepared = False

    def prepare(self):
        """Derived weight copies (folded LayerNorm, transposes, conv packs, dense
        bias images).  Must be re-run whenever parameters change."""
        self.prepared_tail = False
        if self.ws.use_bx3:
            self._prepare_bx3()
        else:
            self._prepare_f32()
        if self.ci != 1:
            self._prepare_edges(net_dev(self.net))
        self.prepared = True

    def _prepare_edges(self, dev):
        """in_chans != 1 (RGB, network_swinir.py:722-727): conv_first and conv_last run on the exact-f32 3x3 conv kernel
        with the image channels zero-padded to 4 (its Cin granularity); the padded weights are zero, so the padded channel
        carries nothing forward or backward."""
        net, C, D, ci = self.net, self.C, self.derived, self.ci
        w4 = D.get("first.w4", C, 4, 3, 3, device=dev)
        w4.zero_()
        w4[:, :ci].copy_(net.conv_first.weight.data)
        ops.pack_conv_weight(w4, D.get("first.wp", 9, C, 4, device=dev), D.get("first.wpt", 9, 4, C, device=dev))
        if not self.direct:
            nf = net.num_feat
            w4 = D.get("last.w4", 4, nf, 3, 3, device=dev)
            b4 = D.get("last.b4", 4, device=dev)
            w4.zero_()
            b4.zero_()
            w4[:ci].copy_(net.conv_last.weight.data)
            b4[:ci].copy_(net.conv_last.bias.data)
            ops.pack_conv_weight(w4, D.get("last.wp", 9, 4, nf, device=dev), D.get("last.wpt", 9, nf, 4, device=dev))

    def _prepare_bx3(self):
        """bf16x3 planes of every matmul operand + folded biases + bias images, by ONE
        launch over a job table built once (rebuilt if a parameter moved)."""
        net, C, hid, D, ws = self.net, self.C, self.hid, self.derived, self.ws
        sig = tuple(p.data_ptr() for p in net.parameters())
        if self.conv3:
            self._refresh_conv3(net.conv_first.weight.device)
        if self._prep is None or sig != self._prep_sig:
            dev = net.conv_first.weight.device
            tb = ops.PrepTable()
            for i, b in enumerate(self.blocks):
                g1, g2 = b.norm1.weight.data, b.norm2.weight.data
                wq, wp = b.attn.qkv.weight.data, b.attn.proj.weight.data
                w1, w2 = b.mlp.fc1.weight.data, b.mlp.fc2.weight.data
                tb.linear(wq, ws.planes(f"{i}.wq", 3 * C, C, dev), gamma=g1)
                tb.linear(wq, ws.planes(f"{i}.wqT", C, 3 * C, dev), gamma=g1, transpose=True)
                tb.linear(wp, ws.planes(f"{i}.wproj", C, C, dev))
                tb.linear(wp, ws.planes(f"{i}.wpT", C, C, dev), transpose=True)
                tb.linear(w1, ws.planes(f"{i}.w1", hid, C, dev), gamma=g2)
                tb.linear(w1, ws.planes(f"{i}.w1T", C, hid, dev), gamma=g2, transpose=True)
                tb.linear(w2, ws.planes(f"{i}.w2", C, hid, dev))
                tb.linear(w2, ws.planes(f"{i}.w2T", hid, C, dev), transpose=True)
                tb.fold_bias(wq, self._qkv_bias(b), b.norm1.bias.data, D.get(f"{i}.bq", 3 * C, device=dev))
                tb.fold_bias(w1, b.mlp.fc1.bias.data, b.norm2.bias.data, D.get(f"{i}.b1", hid, device=dev))
                tb.bias_expand(b.attn.relative_position_bias_table.data,
                               D.get(f"{i}.biasT", b.num_heads, 64, 64, device=dev),
                               D.get(f"{i}.biasN", b.num_heads, 64, 64, device=dev), b.num_heads,
                               biasF=D.get(f"{i}.biasF", b.num_heads, 64, 64, device=dev)
                               if ops.wattn_f16_ok(C, b.num_heads) else None,
                               biasG=D.get(f"{i}.biasG", b.num_heads, 64, 64, device=dev)
                               if ops.wattn_f16_ok(C, b.num_heads) else None)
            for name, conv in self._convs():
                co, ci = conv.weight.shape[:2]
                tb.conv(conv.weight.data, ws.planes(name + ".wp", 9 * co, ci, dev))
                tb.conv(conv.weight.data, ws.planes(name + ".wpt", 9 * ci, co, dev), data_grad=True)
            if self.conv3:
                P = self.c4p
                for name, _, _ in self._rconvs():
                    tb.linear(D.d[name + ".w1p"], ws.planes(name + ".1.w", P, P, dev))
                    tb.linear(D.d[name + ".w1p"], ws.planes(name + ".1.wT", P, P, dev), transpose=True)
            self._prep, self._prep_sig = tb.build(dev), sig
        self._prep.run()

    def _prepare_f32(self):
        net, C, hid, D = self.net, self.C, self.hid, self.derived
        dev = net.conv_first.weight.device
        if self.conv3:
            self._refresh_conv3(dev)
        for i, b in enumerate(self.blocks):
            heads = b.num_heads
            wq = D.get(f"{i}.wq", 3 * C, C, device=dev)
            bq = D.get(f"{i}.bq", 3 * C, device=dev)
            ops.fold_layernorm(b.attn.qkv.weight.data, self._qkv_bias(b), b.norm1.weight.data,
                               b.norm1.bias.data, wq, bq)
            ops.transpose(wq, D.get(f"{i}.wqT", C, 3 * C, device=dev))
            ops.transpose(b.attn.proj.weight.data, D.get(f"{i}.wpT", C, C, device=dev))
            w1 = D.get(f"{i}.w1", hid, C, device=dev)
            b1 = D.get(f"{i}.b1", hid, device=dev)
            ops.fold_layernorm(b.mlp.fc1.weight.data, b.mlp.fc1.bias.data, b.norm2.weight.data,
                               b.norm2.bias.data, w1, b1)
            ops.transpose(w1, D.get(f"{i}.w1T", C, hid, device=dev))
            ops.transpose(b.mlp.fc2.weight.data, D.get(f"{i}.w2T", hid, C, device=dev))
            ops.bias_expand(b.attn.relative_position_bias_table.data,
                            D.get(f"{i}.biasT", heads, 64, 64, device=dev),
                            D.get(f"{i}.biasN", heads, 64, 64, device=dev))
        for name, conv in self._convs():
            co, ci = conv.weight.shape[:2]
            ops.pack_conv_weight(conv.weight.data, D.get(name + ".wp", 9, co, ci, device=dev),
                                 D.get(name + ".wpt", 9, ci, co, device=dev))
        ws = self.ws
        for i, b in enumerate(self.blocks):
            for k in ("wq", "wqT", "wpT", "w1", "w1T", "w2T"):
                ws.register(f"{i}.{k}", D.d[f"{i}.{k}"])
            ws.register(f"{i}.wproj", b.attn.proj.weight.data)
            ws.register(f"{i}.w2", b.mlp.fc2.weight.data)
        for name, _ in self._convs():
            ws.register(name + ".wp", D.d[name + ".wp"])
            ws.register(name + ".wpt", D.d[name + ".wpt"])
        if self.conv3:
            P = self.c4p
            for name, _, _ in self._rconvs():
                ops.transpose(D.d[name + ".w1p"], D.get(name + ".1.wT", P, P, device=dev))
                ws.register(name + ".1.w", D.d[name + ".w1p"])
                ws.register(name + ".1.wT", D.d[name + ".1.wT"])

    def _rconvs(self):
        """(name, module, parameter prefix) of the convs in front of the residual connections."""
        for li, layer in enumerate(self.net.layers):
            yield f"l{li}", layer.conv, f"layers.{li}.conv."
        yield "cab", self.net.conv_after_body, "conv_after_body."

    def _refresh_conv3(self, dev):
        """'3conv': zero-padded copies of the three convs' parameters (read by the weight-preparation table)."""
        D, c4, P, C = self.derived, self.c4, self.c4p, self.C
        for name, mod, _ in self._rconvs():
            fresh = name + ".w0p" not in D.d
            w0, b0 = D.get(name + ".w0p", P, C, 3, 3, device=dev), D.get(name + ".b0p", P, device=dev)
            w1, b1 = D.get(name + ".w1p", P, P, device=dev), D.get(name + ".b1p", P, device=dev)
            w4 = D.get(name + ".w4p", C, P, 3, 3, device=dev)
            if fresh:
                for t in (w0, b0, w1, b1, w4):
                    t.zero_()
            w0[:c4].copy_(mod[0].weight.data)
            b0[:c4].copy_(mod[0].bias.data)
            w1[:c4, :c4].copy_(mod[2].weight.data.view(c4, c4))
            b1[:c4].copy_(mod[2].bias.data)
            w4[:, :c4].copy_(mod[4].weight.data)

    def _convs(self):
        net = self.net
        if self.conv3:
            class _W:       # a 3x3 conv whose (padded) weight lives in the derived set
                def __init__(self, w):
                    self.weight = w
            for name, _, _ in self._rconvs():
                yield name + ".0", _W(self.derived.d[name + ".w0p"])
                yield name + ".4", _W(self.derived.d[name + ".w4p"])
        else:
            for name, mod, _ in self._rconvs():
                yield name, mod
        if self.direct:
            yield "up", net.upsample[0]
        elif self.nearest:
            yield "cbu", net.conv_before_upsample[0]
            yield "nup1", net.conv_up1
            yield "nup2", net.conv_up2
            yield "nhr", net.conv_hr
        else:
            yield "cbu", net.conv_before_upsample[0]
            for i in range(self.stages):
                yield f"up{i}", net.upsample[2 * i]

    def _eval_tail_ok(self):
        nf = self.net.num_feat
        return (self.ws.use_bx3 and not self.direct and not self.nearest and self.stages > 0 and self.net.in_chans == 1
                and (self.scale & (self.scale - 1)) == 0
                and nf % 64 == 0 and ops.ps2_fusable(nf, 4 * nf) and ops.F16X2_CONV and ops.F16X2_CONV_WIDE)

    def _eval_tail_planes(self, dev):
        """the 'pixelshuffle' upsampler's convs in sub-pixel-major column order (PrepTable.conv(ps2=True)): evaluation only"""
        if getattr(self, "_tail_planes", None) is None or not self.prepared_tail:
            tb = ops.PrepTable()
            nf = self.net.num_feat
            self._tail_planes = []
            for i in range(self.stages):
                wp = ops.Bx3(9 * 4 * nf, nf, dev)
                tb.conv(self.net.upsample[2 * i].weight.data, wp, ps2=True)
                self._tail_planes.append(wp)
            tb.build(dev).run()
            self.prepared_tail = True
        return self._tail_planes

    # ------------------------------------------------------------------ forward
    def forward(self, x, dp=None, save=True):
        """x: [B,H,W] fp32 cuda (in_chans 1) or NHWC [B,H,W,4] (in_chans 2..4, zero-padded), H and W multiples of 8
        -> [B,in_chans,s*H,s*W].  dp: None or a [2*nblocks, B] tensor of DropPath multipliers."""
        if not self.prepared:
            self.prepare()
        net, C, hid, D, ws = self.net, self.C, self.hid, self.derived, self.ws
        B, H, W = x.shape[:3]
        assert x.dim() == (3 if self.ci == 1 else 4) and (self.ci == 1 or x.shape[3] == 4), x.shape
        T = B * H * W
        dev = x.device
        bufs = self.bufs
        tag = "t" if save else "e"    # eval shares one set of block buffers

        def buf(name, *shape):
            return bufs.get(f"{tag}.{name}", *shape, device=dev)

        def resi_conv(name, mod, key, src, dst, skip):
            """dst = conv(src) + skip, the conv being '1conv' or '3conv'; returns what the backward needs."""
            if not self.conv3:
                ops.conv3x3(src, ws[name + ".wp"], mod.bias.data, C, out=dst, epi=2, R=skip)
                return None
            P = self.c4p
            c0, c1 = buf(key + ".c0", B, H, W, P), buf(key + ".c1", B, H, W, P)
            ops.conv3x3(src, ws[name + ".0.wp"], D.d[name + ".b0p"], P, out=c0, epi=6, alpha=0.2)
            ops.gemm_nt(c0.view(T, P), ws[name + ".1.w"], D.d[name + ".b1p"], out=c1.view(T, P))
            ops.leaky_relu_(c1, 0.2)
            ops.conv3x3(c1, ws[name + ".4.wp"], mod[4].bias.data, C, out=dst, epi=2, R=skip)
            return c0, c1

        f0 = buf("f0", B, H, W, C)
        if self.ci == 1:
            ops.conv3x3_cin1_fwd(x, net.conv_first.weight.data, net.conv_first.bias.data, C, out=f0)
        else:
            ops.conv3x3(x, D.d["first.wp"], net.conv_first.bias.data, C, out=f0)
        st_pe = buf("st_pe", T, 2)
        t = buf("t0", T, C)
        if net.patch_norm:
            ops.layernorm_fwd(f0.view(T, C), st_pe, t, net.patch_embed.norm.weight.data,
                              net.patch_embed.norm.bias.data)
        else:           # patch_norm=False: the tokens are conv_first's output (a copy: the blocks and `ape` write t)
            t.copy_(f0.view(T, C))
        if net.ape:     # network_swinir.py:918-919: one [H*W, C] table, the same for every patch of the batch
            pos = net.absolute_pos_embed.data.view(H * W, C)
            for b in range(B):
                ops.axpby(t[b * H * W:(b + 1) * H * W], pos, 1.0, 1.0)
        sv = {"x": x, "f0": f0, "st_pe": st_pe, "blocks": [], "layers": [], "B": B, "H": H, "W": W,
              "dp": dp}
        bi = 0
        # LayerNorm statistics can come out of the producing GEMM's epilogue (env SRHIP_FUSE_STATS)
        fuse = ws.use_bx3 and os.environ.get("SRHIP_FUSE_STATS", "1") != "0"
        for li, layer in enumerate(net.layers):
            t_in = t
            nblk = len(layer.residual_group.blocks)
            st1 = None                  # statistics of t, if the previous block's fc2 produced them
            for j, blk in enumerate(layer.residual_group.blocks):
                k = bi if save else 0
                heads = blk.num_heads
                s1 = None if dp is None else dp[2 * bi]
                s2 = None if dp is None else dp[2 * bi + 1]
                if st1 is None:
                    st1 = buf(f"{k}.st1", T, 2)
                    ops.layernorm_fwd(t, st1)
                # inference with 5 / 6 heads: the fused kernel keeps q, k, v in registers and writes no qkv at all
                no_qkv = (not save) and self.fuse_wmsa and fuse and heads in (5, 6)
                qkv = None if no_qkv else buf(f"{k}.qkv", T, 3 * C)
                a = buf(f"{k}.a", T, C)
                x1 = buf(f"{k}.x1", T, C)
                st2 = buf(f"{k}.st2", T, 2)
                # (under --amp too: the fused kernels keep their three products -- faster than the separate one-product launches)
                if self.fuse_wmsa and fuse:
                    ops.wmsa_fwd_f16(t, st1, ws[f"{bi}.wq"], D.d[f"{bi}.bq"], ws[f"{bi}.wproj"],
                                     blk.attn.proj.bias.data, D.d[f"{bi}.biasF"], qkv, a, x1, B, H, W, heads,
                                     blk.shift_size, rowscale=s1, stats_out=st2)
                else:
                    ops.gemm_nt(t, ws[f"{bi}.wq"], D.d[f"{bi}.bq"], out=qkv, a_mode=1, ln_stats=st1)
                    if ops.wattn_f16_ok(C, heads):
                        ops.window_attention_fwd_f16(qkv, a, D.d[f"{bi}.biasF"], B, H, W, C, heads, blk.shift_size)
                    else:
                        ops.window_attention_fwd(qkv, a, D.d[f"{bi}.biasT"], B, H, W, C, heads, blk.shift_size)
                    ops.gemm_nt(a, ws[f"{bi}.wproj"], blk.attn.proj.bias.data, out=x1, epi=2, R=t,
                                rowscale=s1, rows_per_scale=H * W, stats_out=st2 if fuse else None)
                    if not fuse:
                        ops.layernorm_fwd(x1, st2)
                # block outputs ping-pong in eval, are kept per block in training
                x2 = buf(f"{bi if save else bi % 2}.x2", T, C)
                st_next = None
                if fuse and j + 1 < nblk:
                    st_next = buf(f"{(bi + 1) if save else (bi + 1) % 2}.st1", T, 2)
                if self.fuse_mlp_h:
                    h = buf(f"{k}.h", T, hid) if save else None       # inference never reads it
                    ops.mlp_fwd_f16(x1, st2, ws[f"{bi}.w1"], D.d[f"{bi}.b1"], ws[f"{bi}.w2"], blk.mlp.fc2.bias.data,
                                    x2, h=h, rowscale=s2, rows_per_scale=H * W, stats_out=st_next)
                else:
                    h = buf(f"{k}.h", T, hid)
                    ops.gemm_nt(x1, ws[f"{bi}.w1"], D.d[f"{bi}.b1"], out=h, a_mode=1, ln_stats=st2)
                    ops.gemm_nt(h, ws[f"{bi}.w2"], blk.mlp.fc2.bias.data, out=x2, a_mode=2, epi=2,
                                R=x1, rowscale=s2, rows_per_scale=H * W, stats_out=st_next)
                if save:
                    sv["blocks"].append((t, st1, qkv, a, x1, st2, h))
                t = x2
                st1 = st_next
                bi += 1
            tl = buf(f"L{li if save else li % 2}.out", T, C)
            rc = resi_conv(f"l{li}", layer.conv, f"L{li if save else 0}", t.view(B, H, W, C), tl.view(B, H, W, C),
                           t_in.view(B, H, W, C))
            if save:
                sv["layers"].append((t_in, t, rc))
            t = tl
        st_n = buf("st_n", T, 2)
        tn = buf("tn", T, C)
        ops.layernorm_fwd(t, st_n, tn, net.norm.weight.data, net.norm.bias.data)
        f = buf("f", B, H, W, C)
        sv["cab_rc"] = resi_conv("cab", net.conv_after_body, "cab", tn.view(B, H, W, C), f, f0)
        r = self.scale
        y = torch.empty(B, net.in_chans, H * r, W * r, device=dev) if not save else \
            buf("y", B, net.in_chans, H * r, W * r)
        def conv_last(src, h, w):
            if self.ci == 1:
                ops.conv3x3_cout1_fwd(src, net.conv_last.weight.data, net.conv_last.bias.data, out=y.view(B, h, w))
                return
            o4 = ops.conv3x3(src, D.d["last.wp"], D.d["last.b4"], 4, out=buf("last.o4", B, h, w, 4))
            y.copy_(o4[..., :self.ci].permute(0, 3, 1, 2))      # NHWC (padded to 4) -> NCHW: data movement

        if self.direct:     # conv 180 -> s*s, PixelShuffle(s) (network_swinir.py:943-947)
            cu = r * r * net.in_chans
            u = buf("u", B, H, W, cu)
            ops.conv3x3(f, ws["up.wp"], net.upsample[0].bias.data, cu, out=u)
            ops.pixel_shuffle(u, r, out=y)
        elif self.nearest:  # 'nearest_conv' (network_swinir.py:948-961)
            nf = net.num_feat
            u = buf("cbu", B, H, W, nf)
            ops.conv3x3(f, ws["cbu.wp"], net.conv_before_upsample[0].bias.data, nf, out=u, epi=6, alpha=0.01)
            n1 = ops.nearest_up2(u, buf("n1", B, 2 * H, 2 * W, nf))
            a1 = buf("na1", B, 2 * H, 2 * W, nf)
            ops.conv3x3(n1, ws["nup1.wp"], net.conv_up1.bias.data, nf, out=a1, epi=6, alpha=0.2)
            n2 = ops.nearest_up2(a1, buf("n2", B, 4 * H, 4 * W, nf))
            a2 = buf("na2", B, 4 * H, 4 * W, nf)
            ops.conv3x3(n2, ws["nup2.wp"], net.conv_up2.bias.data, nf, out=a2, epi=6, alpha=0.2)
            a3 = buf("na3", B, 4 * H, 4 * W, nf)
            ops.conv3x3(a2, ws["nhr.wp"], net.conv_hr.bias.data, nf, out=a3, epi=6, alpha=0.2)
            conv_last(a3, 4 * H, 4 * W)
            if save:
                sv["near"] = (u, n1, a1, n2, a2, a3)
        elif not save and self._eval_tail_ok():
            # evaluation: the upsampler's convs store through their PixelShuffle(2) (no separate shuffle pass); under --amp on
            # fp16 storage (conv_h16.hip: the 64-channel maps at up to 256 x 256 are what the tail's time goes into)
            nf = net.num_feat
            u = buf("cbu", B, H, W, nf)
            ops.conv3x3(f, ws["cbu.wp"], net.conv_before_upsample[0].bias.data, nf, out=u, epi=6, alpha=0.01)
            ups_w = self._eval_tail_planes(dev)
            h, w = H, W
            if ops.h16_eval():
                u = u.half()
                for i in range(self.stages):
                    u = ops.conv3x3_h16(u, ups_w[i], net.upsample[2 * i].bias.data, 4 * nf,
                                        out=bufs.get(f"h.upu{i}", B, 2 * h, 2 * w, nf, device=dev, dtype=torch.float16), ps2=True)
                    h, w = 2 * h, 2 * w
                ops.conv3x3_cout1_h16(u, net.conv_last.weight.data, net.conv_last.bias.data, out=y.view(B, h, w))
            else:
                for i in range(self.stages):
                    un = buf(f"upu{i}", B, 2 * h, 2 * w, nf)
                    ops.conv3x3_ps2(u, ups_w[i], net.upsample[2 * i].bias.data, un)
                    u, h, w = un, 2 * h, 2 * w
                ops.conv3x3_cout1_fwd(u, net.conv_last.weight.data, net.conv_last.bias.data, out=y.view(B, h, w))
        else:               # 'pixelshuffle' (network_swinir.py:937-942)
            nf = net.num_feat
            u = buf("cbu", B, H, W, nf)
            ops.conv3x3(f, ws["cbu.wp"], net.conv_before_upsample[0].bias.data, nf, out=u, epi=6, alpha=0.01)
            h, w, ups = H, W, [u]
            for i in range(self.stages):
                c = buf(f"upc{i}", B, h, w, 4 * nf)
                ops.conv3x3(u, ws[f"up{i}.wp"], net.upsample[2 * i].bias.data, 4 * nf, out=c)
                u = buf(f"upu{i}", B, 2 * h, 2 * w, nf)
                ops.pixel_shuffle(c, 2, nhwc_out=True, out=u)
                h, w = 2 * h, 2 * w
                ups.append(u)
            conv_last(u, h, w)
            if save:
                sv["ups"] = ups
        if save:
            sv.update(t_last=t, st_n=st_n, tn=tn, f=f)
            self.saved = sv
        return y

    # ------------------------------------------------------------------ backward
    def _bias_table_grads(self, layer, pre, dparts, dbT_all, bi0, j_lo, j_hi, lheads, B, H, W, G):
        """Relative-position-bias table gradients of the layer's blocks [j_lo, j_hi) (bi0 = global index of block 0 of
        the layer): the partial tiles of the attention backward reduced, then one batched launch per <= 8 blocks."""
        if j_hi <= j_lo:
            return
        tabs = [G(pre + f"residual_group.blocks.{j}.attn.relative_position_bias_table") for j in range(j_lo, j_hi)]
        same = len(lheads) == 1
        if dparts is not None:
            ops.wattn_dbias_reduce_f16(dparts[j_lo:j_hi], dbT_all, bi0 + j_lo, B, H, W, next(iter(lheads)))
        for j0 in range(0, j_hi - j_lo, 8):
            if same:
                ops.bias_grad_batched(dbT_all, bi0 + j_lo + j0, tabs[j0:j0 + 8])
            else:
                for j in range(j_lo + j0, min(j_hi, j_lo + j0 + 8)):
                    ops.bias_grad(dbT_all[bi0 + j, :layer.residual_group.blocks[j].num_heads], tabs[j - j_lo])

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        """dy: [B,in_chans,s*H,s*W]; grads: dict name -> tensor to receive the parameter
        gradient (overwritten).  Returns d loss / d x in the layout forward() took if need_dx."""
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        net, C, hid, D, ws = self.net, self.C, self.hid, self.derived, self.ws
        B, H, W, dp = sv["B"], sv["H"], sv["W"], sv["dp"]
        T = B * H * W
        dev = dy.device
        bufs = self.bufs
        r = self.scale
        cu = r * r * net.in_chans

        def buf(name, *shape):
            return bufs.get("g." + name, *shape, device=dev)

        def G(name):
            return grads[name]

        def conv_last_bwd(src, h, w, gsrc):
            """conv_last: parameter gradients, and gsrc = d loss / d src"""
            if self.ci == 1:    # 64 -> 1: the 1-channel kernels with the roles of x / dy swapped and flipped taps
                dyv = dy.reshape(B, h, w).contiguous()
                ops.conv3x3_cin1_wgrad(dyv, src, G("conv_last.weight"), None, flip=True)
                ops.sum_into(dyv, G("conv_last.bias"))
                ops.conv3x3_cin1_fwd(dyv, net.conv_last.weight.data, None, net.num_feat, out=gsrc, flip=True)
                return
            g4 = buf("last.g4", B, h, w, 4)         # NCHW -> NHWC padded to 4 channels: data movement
            g4.zero_()
            g4[..., :self.ci].copy_(dy.permute(0, 2, 3, 1))
            dw4, db4 = buf("last.dw4", 4, net.num_feat, 3, 3), buf("last.db4", 4)
            ops.conv3x3_wgrad(g4, src, dw4, db4)
            G("conv_last.weight").copy_(dw4[:self.ci])
            G("conv_last.bias").copy_(db4[:self.ci])
            ops.conv3x3(g4, D.d["last.wpt"], None, net.num_feat, out=gsrc)

        df = buf("df", B, H, W, C)
        if self.direct:
            du = buf("du", B, H, W, cu)
            ops.pixel_shuffle(dy, r, inverse=True, out=du)
            ops.conv3x3_wgrad(du, sv["f"], G("upsample.0.weight"), G("upsample.0.bias"))
            ops.conv3x3(du, ws["up.wpt"], None, C, out=df)
        elif self.nearest:
            nf = net.num_feat
            u, n1, a1, n2, a2, a3 = sv["near"]
            h, w = 4 * H, 4 * W
            g3 = buf("ng3", B, h, w, nf)
            conv_last_bwd(a3, h, w, g3)
            ops.leaky_relu_mask(g3, a3, 0.2)
            ops.conv3x3_wgrad(g3, a2, G("conv_hr.weight"), G("conv_hr.bias"))
            g2 = buf("ng2", B, h, w, nf)
            ops.conv3x3(g3, ws["nhr.wpt"], None, nf, out=g2, epi=7, R=a2, alpha=0.2)       # * LeakyReLU'(conv_up2 output)
            ops.conv3x3_wgrad(g2, n2, G("conv_up2.weight"), G("conv_up2.bias"))
            ops.conv3x3(g2, ws["nup2.wpt"], None, nf, out=g3)                               # d / d(upsampled a1)
            g1 = buf("ng1", B, 2 * H, 2 * W, nf)
            ops.nearest_up2(g1, g3, adjoint=True)
            ops.leaky_relu_mask(g1, a1, 0.2)
            ops.conv3x3_wgrad(g1, n1, G("conv_up1.weight"), G("conv_up1.bias"))
            gn = buf("ngn", B, 2 * H, 2 * W, nf)
            ops.conv3x3(g1, ws["nup1.wpt"], None, nf, out=gn)
            g0 = buf("ng0", B, H, W, nf)
            ops.nearest_up2(g0, gn, adjoint=True)
            ops.leaky_relu_mask(g0, u, 0.01)
            ops.conv3x3_wgrad(g0, sv["f"], G("conv_before_upsample.0.weight"), G("conv_before_upsample.0.bias"))
            ops.conv3x3(g0, ws["cbu.wpt"], None, C, out=df)
        else:
            nf, ups = net.num_feat, sv["ups"]
            h, w = H * r, W * r
            g = buf(f"dupu{self.stages}", B, h, w, nf)
            conv_last_bwd(ups[-1], h, w, g)
            for i in reversed(range(self.stages)):
                h, w = h // 2, w // 2
                dc = buf(f"dupc{i}", B, h, w, 4 * nf)
                ops.pixel_shuffle(g, 2, nhwc_out=True, inverse=True, out=dc)
                ops.conv3x3_wgrad(dc, ups[i], G(f"upsample.{2 * i}.weight"), G(f"upsample.{2 * i}.bias"))
                g = buf(f"dupu{i}", B, h, w, nf)
                if i == 0:      # LeakyReLU backward rides in the data-gradient conv: * (cbu > 0 ? 1 : 0.01)
                    ops.conv3x3(dc, ws["up0.wpt"], None, nf, out=g, epi=7, R=ups[0], alpha=0.01)
                else:
                    ops.conv3x3(dc, ws[f"up{i}.wpt"], None, nf, out=g)
            ops.conv3x3_wgrad(g, sv["f"], G("conv_before_upsample.0.weight"), G("conv_before_upsample.0.bias"))
            ops.conv3x3(g, ws["cbu.wpt"], None, C, out=df)
        def resi_conv_bwd(name, pre, d, src, rc, dsrc):
            """d = gradient of the conv's output; src its input; writes dsrc and the parameter gradients."""
            if not self.conv3:
                ev = side(lambda: ops.conv3x3_wgrad(d, src, G(pre + "weight"), G(pre + "bias")))
                ops.conv3x3(d, ws[name + ".wpt"], None, C, out=dsrc)
                return ev
            P, c4 = self.c4p, self.c4
            c0, c1 = rc
            dw4 = buf("rc.dw4", C, P, 3, 3)
            ops.conv3x3_wgrad(d, c1, dw4, G(pre + "4.bias"))
            G(pre + "4.weight").copy_(dw4[:, :c4])
            dc1 = buf("rc.dc1", B, H, W, P)
            ops.conv3x3(d, ws[name + ".4.wpt"], None, P, out=dc1, epi=7, R=c1, alpha=0.2)
            dw1, db1 = buf("rc.dw1", P, P), buf("rc.db1", P)
            ops.linear_wgrad(dc1.view(T, P), c0.view(T, P), dw1, db1)
            G(pre + "2.weight").view(c4, c4).copy_(dw1[:c4, :c4])
            G(pre + "2.bias").copy_(db1[:c4])
            dc0 = buf("rc.dc0", B, H, W, P)
            ops.gemm_nt(dc1.view(T, P), ws[name + ".1.wT"], out=dc0.view(T, P))
            ops.leaky_relu_mask(dc0, c0, 0.2)
            dw0, db0 = buf("rc.dw0", P, C, 3, 3), buf("rc.db0", P)
            ops.conv3x3_wgrad(dc0, src, dw0, db0)
            G(pre + "0.weight").copy_(dw0[:c4])
            G(pre + "0.bias").copy_(db0[:c4])
            ops.conv3x3(dc0, ws[name + ".0.wpt"], None, C, out=dsrc)

        # The weight gradients of ALL blocks of an RSTB layer go through ONE grouped launch + ONE reducer at the layer's
        # end (bf16x3 kernels: up to 24 problems): 4 x depth problems = 48 tiles fill the chip with 5 reduce slices
        # where one block's 8 tiles need 32 -- a sixth of the partial-sum traffic (33 MB written + read per block) and of
        # the reducer's work.  Their operands therefore live until the layer's end: per block its own dh / gelu(h) /
        # dqkv and gradient buffers (1.3 GB per layer at B = 8; sized for 288 GB of HBM).  SRHIP_WGRAD_PER_BLOCK=1 or
        # the exact-f32 kernels: one launch per block, three rotating gradient buffers.
        max_nb = max(len(l.residual_group.blocks) for l in net.layers)
        defer = ws.use_bx3 and 4 * max_nb <= 24 and os.environ.get("SRHIP_WGRAD_PER_BLOCK", "0") != "1"
        nset = max_nb if defer else 1
        # SRHIP_SWIN_SIDE_WGRAD=1 (round 6 experiment): a layer's grouped weight-gradient launch, its reducers, the bias-table
        # reductions and the RSTB conv's weight gradient go to a SIDE stream and run beside the next layer's data-gradient
        # chain (no weight gradient is on that chain).  Their operands then outlive the layer: two sets of the per-layer
        # buffers, by layer parity (+1.3 GB).
        side_on = defer and dev.type == "cuda" and os.environ.get("SRHIP_SWIN_SIDE_WGRAD", "1") == "1"
        if side_on and getattr(self, "wstream", None) is None:
            self.wstream = torch.cuda.Stream(device=dev)
        main_stream = torch.cuda.current_stream() if dev.type == "cuda" else None

        side_done = []       # one event per finished layer on the side stream

        def side(fn):
            if not side_on:
                fn()
                return None
            ev = torch.cuda.Event()
            ev.record(main_stream)
            self.wstream.wait_event(ev)
            with torch.cuda.stream(self.wstream):
                fn()
                done = torch.cuda.Event()
                done.record(self.wstream)
            return done

        dtn = buf("dtn", T, C)
        resi_conv_bwd("cab", "conv_after_body.", df, sv["tn"].view(B, H, W, C), sv["cab_rc"], dtn.view(B, H, W, C))
        dt = buf("dt", T, C)
        ops.layernorm_bwd(dtn, sv["t_last"], sv["st_n"], dt, gamma=net.norm.weight.data,
                          dgamma=G("norm.weight"), dbeta=G("norm.bias"))
        def layer_bufs(par):
            sfx = f".p{par}" if side_on else ""
            return ([buf(f"g{i}{sfx}", T, C) for i in range(2 * nset + 1)], [buf(f"dh{i}{sfx}", T, hid) for i in range(nset)],
                    [buf(f"dqkv{i}{sfx}", T, 3 * C) for i in range(nset)], sfx)
        gbufs, dhs, dqkvs, sfx = layer_bufs(0)
        # gelu(h), operand of the fc2 weight gradient: by-product of the dgelu epilogue.  SRHIP_RECOMPUTE_GH=1 (round 6, VERDICT
        # r5 item 1a): the fused backward does NOT store it (47 MB per block less written at B = 8) and the grouped weight-gradient
        # launch recomputes it from the saved h in its operand prologue (b_mode 2, the forward's own packed x Phi(x)).  Measured
        # on the README step, same box, two rounds each: the MLP backward 127 -> 112 us per launch (-0.35 ms per step), the
        # grouped launch +160 us each (the GELU sits on the two B-operand staging waves of the fc2 tiles, the launch is ONE
        # round of 240 blocks and ends with its slowest block): 689 against 708 patches/s -- a loss, so it stays opt-in.
        recompute_gh = (self.fuse_mlp_h and ws.use_bx3 and ops.F16X2 and os.environ.get("SRHIP_RECOMPUTE_GH", "0") == "1")
        ghs = [None if recompute_gh else buf(f"gh{i}", T, hid) for i in range(nset)]
        ghs2 = [None if recompute_gh else buf(f"gh{i}.p1", T, hid) for i in range(nset)] if side_on else ghs
        dxh, da = buf("dxh", T, C), buf("da", T, C)
        nrot = len(gbufs)
        bi = len(self.blocks)
        # bias-gradient images of all blocks (each overwritten by its attention backward: no memset)
        dbT_all = buf("dbiasT_all", len(self.blocks), max(b.num_heads for b in self.blocks), 64, 64)
        for li in reversed(range(len(net.layers))):
            layer = net.layers[li]
            t_in, t_blocks, rc = sv["layers"][li]
            pre = f"layers.{li}."
            if side_on:        # this layer's set: the other one may still be read by the layer behind it on the side stream
                gbufs, dhs, dqkvs, sfx = layer_bufs(li & 1)
                if len(side_done) >= 2:          # ... and the layer two back, which had THIS set, has to be through with it
                    main_stream.wait_event(side_done[-2])
            ghs_l = ghs2 if (side_on and (li & 1)) else ghs
            gi = 0
            g = gbufs[gi]
            conv_done = resi_conv_bwd(f"l{li}", pre + "conv.", dt.view(B, H, W, C), t_blocks.view(B, H, W, C), rc,
                                      g.view(B, H, W, C))
            nb = len(layer.residual_group.blocks)
            # data-parallel runs: layer 0 announces its gradients in up to three buckets (bucket_prefixes); `cuts` maps
            # the block index at which a group is complete to the bucket to announce there
            cuts = {}
            if li == 0 and on_layer_done is not None and len(net.layers) > 1 and defer:
                grp = self.layer0_bucket_groups()
                if grp is not None:
                    cuts[grp[0][0]] = len(net.layers) - 1
                    if grp[1]:
                        cuts[grp[1][0]] = len(net.layers)
            flushed_hi = nb              # blocks [flushed_hi, nb) have their weight / bias-table gradients enqueued
            pending = []                 # weight-gradient problems of the layer's blocks (deferred form)
            front = None                 # a block's qkv data gradient handed to the next block's fused MLP backward
            # partial bias-gradient tiles of the layer's blocks: reduced by ONE launch per layer (same head count only)
            lheads = {blk.num_heads for blk in layer.residual_group.blocks}
            dparts = None
            if len(lheads) == 1 and ops.wattn_f16_ok(C, next(iter(lheads))):
                nbmax = max(len(l.residual_group.blocks) for l in net.layers)
                dparts = buf("dbias_parts" + (f".p{li & 1}" if side_on else ""), nbmax,
                             ops.wattn_dbias_ws(B, H, W, max(b.num_heads for b in self.blocks)))
                dparts = dparts[:nb, :ops.wattn_dbias_ws(B, H, W, next(iter(lheads)))]
            for j in reversed(range(nb)):
                bi -= 1
                dh, gh, dqkv = dhs[j % nset], ghs_l[j % nset], dqkvs[j % nset]
                blk = layer.residual_group.blocks[j]
                p = pre + f"residual_group.blocks.{j}."
                t, st1, qkv, a, x1, st2, h = sv["blocks"][bi]
                heads = blk.num_heads
                s1 = None if dp is None else dp[2 * bi]
                s2 = None if dp is None else dp[2 * bi + 1]
                g1, gout = gbufs[(gi + 1) % nrot], gbufs[(gi + 2) % nrot]
                # ---- MLP branch: x2 = x1 + s2*(gelu(h) W2^T + b2)
                chained = self.fuse_mlp_h and self.chain_proj and getattr(ws[f"{bi}.wpT"], "fmt", 0) == 1
                if self.fuse_mlp_h:
                    # `front`: the qkv Linear's data gradient + LayerNorm backward of the block behind this one (its
                    # result is this MLP's incoming gradient g) runs as the first phase of the same kernel
                    ops.mlp_bwd_f16(g, ws[f"{bi}.w2T"], ws[f"{bi}.w1T"], h, dh, gh, x1, st2, g1, rowscale=s2,
                                    rows_per_scale=H * W,
                                    chain=(ws[f"{bi}.wpT"], da, s1) if chained else None, front=front)
                    front = None
                else:
                    ops.gemm_nt(g, ws[f"{bi}.w2T"], None, out=dh, epi=3, R=h, rowscale=s2,
                                rows_per_scale=H * W, aux=gh)
                    if ws.use_bx3:      # LayerNorm backward fused into the GEMM epilogue
                        ops.gemm_nt_lnbwd(dh, ws[f"{bi}.w1T"], x1, st2, g, g1)
                    else:
                        ops.gemm_nt(dh, ws[f"{bi}.w1T"], None, out=dxh)
                        ops.layernorm_bwd(dxh, x1, st2, g1, res=g)
                # ---- attention branch: x1 = t + s1*(a Wp^T + bp)
                if not chained:
                    ops.gemm_nt(g1, ws[f"{bi}.wpT"], None, out=da, epi=2, rowscale=s1, rows_per_scale=H * W)
                dbT = dbT_all[bi, :heads]
                if ops.wattn_f16_ok(C, heads):
                    ops.window_attention_bwd_f16(qkv, da, dqkv, D.d[f"{bi}.biasF"], D.d[f"{bi}.biasG"], dbT, B, H,
                                                 W, C, heads, blk.shift_size,
                                                 parts=None if dparts is None else dparts[j])
                else:
                    ops.window_attention_bwd(qkv, da, dqkv, D.d[f"{bi}.biasT"], D.d[f"{bi}.biasN"], dbT, B, H,
                                             W, C, heads, blk.shift_size)
                if (ws.use_bx3 and self.fuse_mlp_h and self.front_qkv and j > 0
                        and getattr(ws[f"{bi}.wqT"], "fmt", 0) == 1 and not ops.lib.srhip_get_matmul_mode()):
                    front = (dqkv, ws[f"{bi}.wqT"], t, st1, g1)       # -> gout, inside the next block's MLP backward
                elif ws.use_bx3:
                    ops.gemm_nt_lnbwd(dqkv, ws[f"{bi}.wqT"], t, st1, g1, gout)
                else:
                    ops.gemm_nt(dqkv, ws[f"{bi}.wqT"], None, out=dxh)
                    ops.layernorm_bwd(dxh, t, st1, gout, res=g1)
                # ---- the four weight gradients of the block: one launch per block, or collected for the layer's
                problems = [
                    dict(dY=dqkv, X=t, dW=G(p + "attn.qkv.weight"),
                         db=G(p + "attn.qkv.bias") if blk.attn.qkv.bias is not None else buf("dbq.unused", 3 * C), b_mode=1,
                         ln_stats=st1, ln=(blk.attn.qkv.weight.data, blk.norm1.weight.data,
                                           blk.norm1.bias.data, G(p + "norm1.weight"), G(p + "norm1.bias"))),
                    dict(dY=g, X=h if gh is None else gh, b_mode=2 if gh is None else 0, dW=G(p + "mlp.fc2.weight"),
                         db=G(p + "mlp.fc2.bias"), a_rowscale=s2, a_rowscale_rows=H * W),
                    dict(dY=dh, X=x1, dW=G(p + "mlp.fc1.weight"), db=G(p + "mlp.fc1.bias"), b_mode=1,
                         ln_stats=st2, ln=(blk.mlp.fc1.weight.data, blk.norm2.weight.data,
                                           blk.norm2.bias.data, G(p + "norm2.weight"), G(p + "norm2.bias"))),
                    dict(dY=g1, X=a, dW=G(p + "attn.proj.weight"), db=G(p + "attn.proj.bias"),
                         a_rowscale=s1, a_rowscale_rows=H * W),
                ]
                if defer:
                    pending += problems
                else:
                    ops.linear_wgrad_grouped(problems)
                gi = (gi + 2) % nrot
                g = gout
                if j in cuts:            # blocks [j, flushed_hi) are complete: their gradients now, their bucket announced
                    def cut_end(pending=pending, flushed_hi=flushed_hi, j=j, bi=bi):
                        ops.linear_wgrad_grouped(pending)
                        self._bias_table_grads(layer, pre, dparts, dbT_all, bi - j, j, flushed_hi, lheads, B, H, W, G)
                        on_layer_done(cuts[j])       # on the side stream: the bucket's all-reduce is ordered behind it
                    side(cut_end)
                    pending = []
                    flushed_hi = j
            def layer_end(pending=pending, layer=layer, pre=pre, dparts=dparts, bi=bi, flushed_hi=flushed_hi, lheads=lheads, li=li,
                          cuts=cuts):
                if pending:
                    ops.linear_wgrad_grouped(pending)
                # relative-position-bias table gradients of the layer's blocks: one launch (<= 8 blocks each)
                self._bias_table_grads(layer, pre, dparts, dbT_all, bi, 0, flushed_hi, lheads, B, H, W, G)
                if on_layer_done is not None and not cuts:   # this layer's gradients are enqueued (all of them on this stream)
                    on_layer_done(len(net.layers) - 1 - li)
            side_done.append(side(layer_end))
            if conv_done is not None:            # the side stream's conv weight gradient reads dt
                main_stream.wait_event(conv_done)
            ops.axpby(dt, g, 1.0, 1.0)   # RSTB skip: t_out = conv(blocks(t_in)) + t_in
        if side_on:
            main_stream.wait_stream(self.wstream)            # every weight gradient is in before anything reads them
        # patch_embed.norm and the conv_after_body skip (f = conv(..) + f0)
        if net.ape:     # d table = sum over the batch of the token gradient
            dpos = G("absolute_pos_embed").view(H * W, C)
            dpos.copy_(dt[:H * W])
            for b in range(1, B):
                ops.axpby(dpos, dt[b * H * W:(b + 1) * H * W], 1.0, 1.0)
        df0 = buf("df0", T, C)
        if net.patch_norm:
            ops.layernorm_bwd(dt, sv["f0"].view(T, C), sv["st_pe"], df0, res=df.view(T, C),
                              gamma=net.patch_embed.norm.weight.data, dgamma=G("patch_embed.norm.weight"),
                              dbeta=G("patch_embed.norm.bias"))
        else:
            df0.copy_(dt)
            ops.axpby(df0, df.view(T, C), 1.0, 1.0)
        if self.ci != 1:
            dw4 = buf("first.dw4", C, 4, 3, 3)
            ops.conv3x3_wgrad(df0.view(B, H, W, C), sv["x"], dw4, G("conv_first.bias"))
            G("conv_first.weight").copy_(dw4[:, :self.ci])
            if need_dx:     # in the layout of the input: NHWC, 4 channels (the padded one receives an exact zero)
                return ops.conv3x3(df0.view(B, H, W, C), D.d["first.wpt"], None, 4)
            return None
        ops.conv3x3_cin1_wgrad(sv["x"], df0.view(B, H, W, C), G("conv_first.weight"),
                               G("conv_first.bias"))
        if need_dx:
            wflip = net.conv_first.weight.data.flip(2, 3).reshape(1, C, 3, 3).contiguous()
            return ops.conv3x3_cout1_fwd(df0.view(B, H, W, C), wflip, None)
        return None
