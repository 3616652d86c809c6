"""MemNet forward / backward as a fixed sequence of libsrhip launches (the plain CNNs reuse the 3x3-conv
kernels; this one adds BatchNorm2d, csrc/bn.hip).

Reference: dlib/models/network_memnet.py:24-179 -- bicubic interpolation of the LR input (clamped); BN-ReLU-conv 1 -> 64;
``num_memory_blocks`` memory blocks; BN-ReLU-conv1x1 64 -> 1; + the interpolated input.  A memory block runs its whole
chain of residual units ``num_residual_blocks`` times (the reference calls the Sequential of ALL units in its loop,
:69-72), keeps every pass's result, and gates the concatenation of those and of all earlier long-term memories through
BN-ReLU-conv1x1.  A residual unit is x + conv(relu(BN(conv(relu(BN(x)))))); no conv has a bias.

On the device (NHWC, T = B*H*W pixels):
  BatchNorm      training: srhip_bn_stats (batch statistics -> coefficients, running statistics updated in place, in the
                 order of application, as the reference's sequential calls do) + srhip_bn_apply (ReLU fused);
                 evaluation: coefficients from the running statistics; a residual unit's two BatchNorms ride in its
                 first conv (the one in front as the BN-ReLU input prologue of srhip_conv3x3_nhwc_split_ex, the one behind
                 folded into the weight), srhip_bn_apply only for the gate and the two ends
  3x3 convs      the split-operand implicit-GEMM conv; the unit's skip connection is its epilogue (epi 2)
  gate 1x1 conv  the bf16x3 NT GEMM on the [T, gate_channels] concatenation (in row chunks below 2 GiB)
  1-channel ends the edge-conv kernels (the 64 -> 1 1x1 conv as a 3x3 with only its centre tap set)
  backward       ReLU masks ride in the data-gradient convs / GEMM (epi 4); srhip_bn_bwd = one reduction + one apply
                 pass with the unit's skip gradient fused; the weight gradients of a memory block's convs (each unit is
                 applied ``num_residual_blocks`` times) in one batched launch, summed per shared weight.
"""
import torch
import torch.nn.functional as F

from . import ops
from .swinir_engine import _Bufs

CH = 64
GEMM_BYTES_MAX = (1 << 31) - 1


class MemNetEngine:
    def __init__(self, net):
        self.net = net
        self.M = len(net.dense_memory_blocks)
        self.R = net.dense_memory_blocks[0].num_residual_blocks
        self.bufs = _Bufs()
        self.derived = _Bufs()
        self.ws = ops.WeightSet()
        if not (self.ws.use_bx3 and ops.bx3_nt_for(CH)):
            raise NotImplementedError("MemNet (libsrhip) runs on the bf16x3 kernels (SRHIP_MM=f32 is not supported)")
        self._prep = self._prep_sig = None
        self.prepared = False
        self._eval_coefs = False
        self.saved = None
        self.bn = {k: m for k, m in net.named_modules() if isinstance(m, torch.nn.BatchNorm2d)}

    def invalidate(self):
        self.prepared = False
        self._eval_coefs = False
        self._h16_ready = False
        self._h16_overflow = False

    def bucket_prefixes(self):
        """0.3-3 M parameters: one gradient bucket."""
        return [["feature_extractor.", "dense_memory_blocks.", "reconstructor."]]

    # ---- names (network_memnet.py:27-34,59-64)
    @staticmethod
    def _unit(i, j):
        return f"dense_memory_blocks.{i}.recursive_unit.{j}.residual_block"

    def _gc(self, i):
        return (self.R + i + 1) * CH

    def prepare(self):
        net, D, ws = self.net, self.derived, self.ws
        dev = net.reconstructor[2].weight.device
        sig = tuple(p.data_ptr() for p in net.parameters())
        rebuild = self._prep is None or sig != self._prep_sig
        if rebuild:
            tb = ops.PrepTable()
            for i in range(self.M):
                mb = net.dense_memory_blocks[i]
                for j in range(self.R):
                    seq = mb.recursive_unit[j].residual_block
                    for c, idx in ((0, 2), (1, 5)):
                        w = seq[idx].weight.data
                        tb.conv(w, ws.planes(f"m{i}.u{j}.c{c}.wp", 9 * CH, CH, dev))
                        tb.conv(w, ws.planes(f"m{i}.u{j}.c{c}.wpt", 9 * CH, CH, dev), data_grad=True)
                gw = mb.gate_unit[2].weight.data.view(CH, self._gc(i))
                tb.linear(gw, ws.planes(f"m{i}.gw", CH, self._gc(i), dev))
                tb.linear(gw, ws.planes(f"m{i}.gwt", self._gc(i), CH, dev), transpose=True)
            self._prep, self._prep_sig = tb.build(dev), sig
        self._prep.run()
        w3 = D.get("rec.w3", 1, CH, 3, 3, device=dev)           # 1x1 conv 64 -> 1 as the centre tap of a 3x3
        w3.zero_()
        w3[:, :, 1, 1].copy_(net.reconstructor[2].weight.data.view(1, CH))
        self.prepared = True

    def _prepare_eval_coefs(self, dev):
        """coef [4, C] = running_mean, 1/sqrt(running_var + eps), gamma * that, beta (nn.BatchNorm2d in eval mode)."""
        for pre, m in self.bn.items():
            c = self.derived.get("coef." + pre, 4, m.num_features, device=dev)
            rstd = torch.rsqrt(m.running_var.float() + m.eps)
            c[0].copy_(m.running_mean)
            c[1].copy_(rstd)
            c[2].copy_(m.weight.data * rstd)
            c[3].copy_(m.bias.data)
        # The BatchNorm BETWEEN a unit's two convs (residual_block.3) is affine at evaluation time and follows conv 0
        # directly: it is folded into that conv -- w'[co] = a[co] w[co], bias'[co] = b[co] with a = gamma rstd, b = beta -
        # mean a -- and the ReLU becomes the conv's epilogue: one pass over the feature map less per unit application.
        tb = ops.PrepTable()
        for i in range(self.M):
            for j in range(self.R):
                seq = self.net.dense_memory_blocks[i].recursive_unit[j].residual_block
                m = seq[3]
                a = (m.weight.data * torch.rsqrt(m.running_var.float() + m.eps)).float()
                wf = self.derived.get(f"m{i}.u{j}.c0.wfold", CH, CH, 3, 3, device=dev)
                wf.copy_(seq[2].weight.data * a.view(CH, 1, 1, 1))
                bf = self.derived.get(f"m{i}.u{j}.c0.bfold", CH, device=dev)
                bf.copy_(m.bias.data - m.running_mean * a)
                tb.conv(wf, self.ws.planes(f"m{i}.u{j}.c0.wpf", 9 * CH, CH, dev))
        self._fold_prep = tb.build(dev)
        self._fold_prep.run()
        self._eval_coefs = True

    def interpolate(self, x):
        """network_memnet.py:127-139: [B,1,h,w] -> [B,s*h,s*w], bicubic (align_corners False), clamped."""
        s = self.net.upscale
        out = F.interpolate(x, size=(s * x.shape[2], s * x.shape[3]), mode='bicubic', align_corners=False)
        return torch.clamp(out, min=0.0, max=1.0)[:, 0].contiguous()

    def _prepare_h16(self, dev):
        """the gate units' 1x1 convs as the centre tap of 3x3 weights, in the fp16x2 conv operand format (conv_h16.hip)"""
        tb = ops.PrepTable()
        for i in range(self.M):
            gc = self._gc(i)
            w3 = self.derived.get(f"m{i}.gw3", CH, gc, 3, 3, device=dev)
            w3.zero_()
            w3[:, :, 1, 1].copy_(self.net.dense_memory_blocks[i].gate_unit[2].weight.data.view(CH, gc))
            tb.conv(w3, self.ws.planes(f"m{i}.gw3p", 9 * CH, gc, dev))
        tb.build(dev).run()
        self._h16_ready = True

    def forward_h16(self, x):
        """--amp evaluation on fp16 storage (conv_h16.hip): float16 feature maps, one fp16 product; every BatchNorm-ReLU is
        the input prologue of the conv behind it (the gate's 1x1 conv runs as the centre tap of a 3x3 operand)."""
        net, D, ws, R = self.net, self.derived, self.ws, self.R
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape
        dev = x.device
        if not self._eval_coefs:
            self._prepare_eval_coefs(dev)
        if not getattr(self, "_h16_ready", False):
            self._prepare_h16(dev)

        def buf(name, *shape):
            return self.bufs.get("h." + name, *shape, device=dev, dtype=torch.float16)
        a0 = self.bufs.get("h.a0", B, H, W, device=dev)
        ops.bn_apply(xi.view(-1, 1), D.d["coef.feature_extractor.0"], a0.view(-1, 1), relu=True)
        f0 = ops.conv3x3_cin1_h16(a0, net.feature_extractor[2].weight.data, None, CH, out=buf("f0", B, H, W, CH))
        longs, out, n = [f0], f0, 0
        for i in range(self.M):
            gc = self._gc(i)
            cat = buf(f"cat{i}", B, H, W, gc)
            for r in range(R):
                for j in range(R):
                    u = self._unit(i, j)
                    a2 = ops.conv3x3_h16(out, ws[f"m{i}.u{j}.c0.wpf"], D.d[f"m{i}.u{j}.c0.bfold"], CH, out=buf("a2", B, H, W, CH),
                                         epi=1, in_bn=D.d["coef." + u + ".0"])
                    out = ops.conv3x3_h16(a2, ws[f"m{i}.u{j}.c1.wp"], None, CH, out=buf(f"o{n % 2}", B, H, W, CH), epi=2, R=out)
                    n += 1
                cat[..., r * CH:(r + 1) * CH].copy_(out)
            for k, t in enumerate(longs):
                cat[..., (R + k) * CH:(R + k + 1) * CH].copy_(t)
            gate = ops.conv3x3_h16(cat, ws[f"m{i}.gw3p"], None, CH, out=buf(f"gate{i}", B, H, W, CH),
                                   in_bn=D.d[f"coef.dense_memory_blocks.{i}.gate_unit.0"], center_only=True)
            longs.append(gate)
            out = gate
        y = ops.conv3x3_cout1_h16(out, D.d["rec.w3"], None, add=xi, in_bn=D.d["coef.reconstructor.0"])
        return y.view(B, 1, H, W)

    def _h16_ok(self):
        return all(self.ws[f"m{i}.u{j}.c{c}.wp"].fmt == 1 for i in range(self.M) for j in range(self.R) for c in (0, 1))

    # ------------------------------------------------------------------ forward
    def forward(self, x, dp=None, save=True):
        """x [B,h,w] (LR) -> [B,1,s*h,s*w].  BatchNorm follows ``net.training``: batch statistics (and the running-
        statistics update) in training mode, running statistics in eval mode."""
        if not self.prepared:
            self.prepare()
        net, D, ws, R = self.net, self.derived, self.ws, self.R
        training = bool(net.training)
        assert not (save and not training), "MemNet (libsrhip): gradients in eval mode (frozen BatchNorm) are not built"
        if training:
            self._eval_coefs = False        # this forward updates the running statistics: the evaluation-time folds are stale
        if not save and not training and ops.h16_eval() and self._h16_ok() and not getattr(self, "_h16_overflow", False):
            # MemNet's memory blocks add 36 residual units per block without a normalisation in between: with untrained
            # statistics the feature maps leave fp16's range (65504).  One finiteness check of the output image per forward
            # (a 200-ms forward: the host round trip is noise); on overflow this net stays on f32 storage.
            y = self.forward_h16(x)
            # under a hipGraph capture (ModelPlain._graph_forward captures the SECOND call of a weights version) a host
            # read is not permitted: the eager call in front of the capture made the check for these weights
            if torch.cuda.is_current_stream_capturing():
                self.last_eval_path = "fp16 storage"
                return y
            if bool(torch.isfinite(y).all()):
                self.last_eval_path = "fp16 storage"
                return y
            self._h16_overflow = True
        xi = self.interpolate(x[:, None])
        B, H, W = xi.shape
        T = B * H * W
        dev = x.device
        if not training and not self._eval_coefs:
            self._prepare_eval_coefs(dev)
        tag = "t" if save else "e"
        applied = {}

        def buf(name, *shape):
            return self.bufs.get(f"{tag}.{name}", *shape, device=dev)

        def scr(key):
            return self.bufs.get(key, B, H, W, CH, device=dev)

        def bn(pre, key, src, dst):
            m = self.bn[pre]
            if training:
                coef = buf("coef." + key, 4, m.num_features)
                ops.bn_stats(src, m.weight.data, m.bias.data, coef, m.running_mean, m.running_var, m.momentum, m.eps)
                applied[pre] = applied.get(pre, 0) + 1
            else:
                coef = D.d["coef." + pre]
            ops.bn_apply(src, coef, dst, relu=True)
            return coef

        a0 = buf("a0", B, H, W)
        coef0 = bn("feature_extractor.0", "fe", xi, a0)
        f0 = buf("f0", B, H, W, CH)
        ops.conv3x3_cin1_fwd(a0, net.feature_extractor[2].weight.data, None, CH, out=f0)
        longs, out = [f0], f0
        sv_blocks = []
        for i in range(self.M):
            gc = self._gc(i)
            cat = buf(f"m{i}.cat", B, H, W, gc)
            apps = []
            n = 0
            pass_in = []                 # training: the input of every pass over the unit chain (what the backward keeps)
            for r in range(R):
                pass_in.append(out)
                for j in range(R):
                    u = self._unit(i, j)
                    # Round 5: a training forward keeps per unit application only the BatchNorm coefficients (and per pass its
                    # result, the next pass's input); a1 / c1 / a2 and the units' outputs inside a pass go through scratch
                    # buffers and are RECOMPUTED in the backward, one pass at a time, from the pass input and those
                    # coefficients (the same kernels on the same inputs: the same bits).  Keeping all four maps of all 36
                    # applications of all 6 blocks was 464 GB at the README batch (B = 8, 512 x 512): beyond the GPU.
                    key = f"m{i}.r{r}.u{j}"
                    # (the scratch maps are the backward's recompute buffers: dead when the forward ends, rewritten there)
                    a1, c1, a2 = scr("g.rc.a1.0"), scr("g.rc.c1.0"), scr("g.rc.a2.0")
                    nxt = buf(f"m{i}.r{r}.out", B, H, W, CH) if (save and j == R - 1) else scr(f"g.rc.x.{1 + n % 2}")
                    if training:
                        k1 = bn(u + ".0", key + ".k1", out, a1)
                        ops.conv3x3(a1, ws[f"m{i}.u{j}.c0.wp"], None, CH, out=c1)
                        k2 = bn(u + ".3", key + ".k2", c1, a2)
                    else:
                        # conv 0 with the BatchNorm behind it folded into its weight (ReLU: its epilogue) and the BatchNorm-
                        # ReLU in front of it applied to the halo tile as it is staged (srhip_conv3x3_nhwc_split_ex): the
                        # unit is two launches and no pass over the feature map besides them
                        wpf = ws[f"m{i}.u{j}.c0.wpf"]
                        k1 = k2 = None
                        if wpf.fmt == 1:
                            ops.conv3x3(out, wpf, D.d[f"m{i}.u{j}.c0.bfold"], CH, out=a2, epi=1, in_bn=D.d["coef." + u + ".0"])
                        else:
                            bn(u + ".0", key + ".k1", out, a1)
                            ops.conv3x3(a1, wpf, D.d[f"m{i}.u{j}.c0.bfold"], CH, out=a2, epi=1)
                    ops.conv3x3(a2, ws[f"m{i}.u{j}.c1.wp"], None, CH, out=nxt, epi=2, R=out)     # + the unit's input
                    if save:
                        apps.append(dict(k1=k1, k2=k2, j=j))
                    out = nxt
                    n += 1
                cat[..., r * CH:(r + 1) * CH].copy_(out)                                       # short-term memory r
            for k, t in enumerate(longs):
                cat[..., (R + k) * CH:(R + k + 1) * CH].copy_(t)                               # long-term memories
            ag = self.bufs.get("g.ag", T * self._gc(self.M - 1), device=dev)[:T * gc].view(T, gc)     # (the backward's buffer)
            kg = bn(f"dense_memory_blocks.{i}.gate_unit.0", f"m{i}.kg", cat.view(T, gc), ag)
            gate = buf(f"m{i}.gate", B, H, W, CH)
            rows = max(1, GEMM_BYTES_MAX // (4 * gc))
            for r0 in range(0, T, rows):
                ops.gemm_nt(ag[r0:r0 + rows], ws[f"m{i}.gw"], out=gate.view(T, CH)[r0:r0 + rows])
            if save:
                sv_blocks.append(dict(apps=apps, cat=cat, kg=kg, pass_in=pass_in))        # (ag is recomputed from cat and kg)
            longs.append(gate)
            out = gate
        ar = buf("ar", B, H, W, CH)
        kr = bn("reconstructor.0", "kr", out, ar)
        y = buf("y", B, H, W) if save else torch.empty(B, H, W, device=dev)
        ops.conv3x3_cout1_fwd(ar, D.d["rec.w3"], None, out=y)
        ops.axpby(y, xi, 1.0, 1.0)                                                             # + the interpolated input
        for pre, cnt in applied.items():
            self.bn[pre].num_batches_tracked.add_(cnt)
        if save:
            self.saved = dict(xi=xi, a0=a0, coef0=coef0, blocks=sv_blocks, last=out, ar=ar, kr=kr, B=B, H=H, W=W)
        return y.view(B, 1, H, W)

    # ------------------------------------------------------------------ backward
    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False):
        sv = self.saved
        assert sv is not None, "backward() without a saved forward"
        assert not need_dx, "MemNet (libsrhip): no gradient through the bicubic interpolation of the input"
        net, D, ws, R = self.net, self.derived, self.ws, self.R
        B, H, W = sv["B"], sv["H"], sv["W"]
        T = B * H * W
        dev = dy.device
        touched = set()

        def buf(name, *shape):
            return self.bufs.get("g." + name, *shape, device=dev)

        def bn_bwd(pre, dz, x, coef, a=None, dx=None, res=None):
            ops.bn_bwd(dz, x, coef, a=a, dx=dx, res=res, dgamma=grads[pre + ".weight"], dbeta=grads[pre + ".bias"],
                       accumulate=pre in touched)
            touched.add(pre)

        dy = dy.reshape(B, H, W).contiguous()
        # ---- reconstructor: y = conv1x1(relu(BN(last))) + xi
        dw3 = buf("dw3", 1, CH, 3, 3)
        ops.conv3x3_cin1_wgrad(dy, sv["ar"], dw3, None, flip=True)
        grads["reconstructor.2.weight"].view(1, CH).copy_(dw3[:, :, 1, 1])
        d_ar = buf("d_ar", B, H, W, CH)
        ops.conv3x3_cin1_fwd(dy, D.d["rec.w3"], None, CH, out=d_ar, flip=True)
        g_long = [None] * (self.M + 1)                 # gradient wrt long-term memory k (f0, gate_0, ...), accumulated
        g_long[self.M] = buf(f"gl{self.M}", B, H, W, CH)
        bn_bwd("reconstructor.0", d_ar, sv["last"], sv["kr"], a=sv["ar"], dx=g_long[self.M])

        def add_long(k, t):
            if g_long[k] is None:
                g_long[k] = buf(f"gl{k}", B, H, W, CH)
                g_long[k].copy_(t)
            else:
                g_long[k].add_(t)

        for i in reversed(range(self.M)):
            sb = sv["blocks"][i]
            gc = self._gc(i)
            mbn = f"dense_memory_blocks.{i}"
            g_gate = g_long[i + 1].view(T, CH)
            # ---- gate unit: gate = relu(BN(cat)) . Wg^T  (relu(BN(cat)) recomputed from the saved coefficients)
            gcmax = self._gc(self.M - 1)         # one allocation at the widest block's size, views for the others
            ag = buf("ag", T * gcmax)[:T * gc].view(T, gc)
            ops.bn_apply(sb["cat"].view(T, gc), sb["kg"], ag, relu=True)
            ops.linear_wgrad(g_gate, ag, grads[mbn + ".gate_unit.2.weight"].view(CH, gc), None)
            dzg = buf("dzg", T * gcmax)[:T * gc].view(T, gc)
            ops.gemm_nt(g_gate, ws[f"m{i}.gwt"], out=dzg, epi=4, R=ag)                          # * (ag > 0)
            g_cat = buf("gcat", T * gcmax)[:T * gc].view(B, H, W, gc)
            bn_bwd(mbn + ".gate_unit.0", dzg, sb["cat"].view(T, gc), sb["kg"], dx=g_cat.view(T, gc))
            for k in range(i + 1):
                add_long(k, g_cat[..., (R + k) * CH:(R + k + 1) * CH])
            # ---- the R passes over the chain of R residual units, last pass first
            g = None
            dwt = buf("dwt", R * R * 2, CH, CH, 3, 3)
            n = R * R
            for r in reversed(range(R)):
                gs = buf(f"gs{r % 2}", B, H, W, CH)                                            # d / d(pass r's result)
                gs.copy_(g_cat[..., r * CH:(r + 1) * CH])
                if g is not None:
                    gs.add_(g)
                g = gs
                # the pass's unit chain again, forward, from its saved input and BatchNorm coefficients (per-unit buffers,
                # shared by every pass of every block: 4 R maps)
                x = sb["pass_in"][r]
                rec = []
                for j in range(R):
                    ap = sb["apps"][r * R + j]
                    a1, c1 = buf(f"rc.a1.{j}", B, H, W, CH), buf(f"rc.c1.{j}", B, H, W, CH)
                    a2 = buf(f"rc.a2.{j}", B, H, W, CH)
                    ops.bn_apply(x, ap["k1"], a1, relu=True)
                    ops.conv3x3(a1, ws[f"m{i}.u{j}.c0.wp"], None, CH, out=c1)
                    ops.bn_apply(c1, ap["k2"], a2, relu=True)
                    rec.append(dict(x=x, a1=a1, c1=c1, a2=a2, k1=ap["k1"], k2=ap["k2"]))
                    if j < R - 1:
                        nxt = buf(f"rc.x.{j + 1}", B, H, W, CH)
                        ops.conv3x3(a2, ws[f"m{i}.u{j}.c1.wp"], None, CH, out=nxt, epi=2, R=x)
                        x = nxt
                items = []
                for j in reversed(range(R)):
                    n -= 1
                    ap = rec[j]
                    u = self._unit(i, j)
                    items.append((g, ap["a2"], dwt[2 * n + 1], None))
                    dz2 = buf("dz", B, H, W, CH)
                    ops.conv3x3(g, ws[f"m{i}.u{j}.c1.wpt"], None, CH, out=dz2, epi=4, R=ap["a2"])
                    dc1 = buf(f"dc{j}", B, H, W, CH)
                    bn_bwd(u + ".3", dz2, ap["c1"], ap["k2"], dx=dc1)
                    items.append((dc1, ap["a1"], dwt[2 * n], None))
                    dz1 = buf("dz", B, H, W, CH)
                    ops.conv3x3(dc1, ws[f"m{i}.u{j}.c0.wpt"], None, CH, out=dz1, epi=4, R=ap["a1"])
                    gx = buf(f"gx{j}", B, H, W, CH)
                    bn_bwd(u + ".0", dz1, ap["x"], ap["k1"], dx=gx, res=g)                     # + the skip connection
                    g = gx
                ops.conv3x3_wgrad_batched(items)     # the pass's twelve weight gradients (its operands die with the pass)
            dwv = dwt.view(R, R, 2, CH, CH, 3, 3)
            for j in range(R):
                u = self._unit(i, j)
                torch.sum(dwv[:, j, 0], 0, out=grads[u + ".2.weight"])
                torch.sum(dwv[:, j, 1], 0, out=grads[u + ".5.weight"])
            add_long(i, g)                                                                     # the block's input
        # ---- feature extractor: f0 = conv(relu(BN(xi)))
        g_f0 = g_long[0]
        ops.conv3x3_cin1_wgrad(sv["a0"], g_f0, grads["feature_extractor.2.weight"], None)
        wf = D.get("fe.wf", 1, CH, 3, 3, device=dev)
        wf.copy_(net.feature_extractor[2].weight.data.flip(2, 3).reshape(1, CH, 3, 3))
        d_a0 = buf("d_a0", B, H, W)
        ops.conv3x3_cout1_fwd(g_f0, wf, None, out=d_a0)
        bn_bwd("feature_extractor.0", d_a0, sv["xi"], sv["coef0"], a=sv["a0"])
        return None
