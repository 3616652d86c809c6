"""ENLCN as a tape graph (reference dlib/models/network_enlcn.py:369-448): head conv; body = ENLCA, n_resblock ResBlocks
(conv-ReLU-conv, x res_scale, + x) with an ENLCA behind every eighth, a conv; long skip; Upsampler (conv F -> 4F +
PixelShuffle(2) per factor of two) and the output conv.  The F -> 4F convs run as four 3x3 convs of F output channels
(the conv kernels take up to 256 output columns) written side by side, their gradients into the matching rows of the
parameter's.  Trains through the tape's derived backward (ENLCA: Tape.enlca)."""
import math

from .tape import TapeEngine


class ENLCNEngine(TapeEngine):
    def _body(self):
        """(index in net.body, kind) in order"""
        net = self.net
        out, i = [], 0
        out.append((i, "enlca")); i += 1
        for b in range(net.n_resblock):
            out.append((i, "res")); i += 1
            if (b + 1) % 8 == 0:
                out.append((i, "enlca")); i += 1
        out.append((i, "conv"))
        return out

    def bank_entries(self):
        net, bank = self.net, self.bank
        F = net.n_feats
        for i, kind in self._body():
            m = net.body[i]
            if kind == "enlca":
                for sub in ("conv_match1", "conv_match2", "conv_assembly"):
                    c = getattr(m, sub)[0]
                    bank.conv(f"body.{i}.{sub}", c.weight, c.bias, "c1")
            elif kind == "res":
                bank.conv(f"body.{i}.0", m.body[0].weight, m.body[0].bias, "c3")
                bank.conv(f"body.{i}.2", m.body[2].weight, m.body[2].bias, "c3")
            else:
                bank.conv(f"body.{i}", m.weight, m.bias, "c3")
        for st in range(int(math.log2(net.upscale))):
            c = net.tail[0][2 * st]
            for j in range(4):
                bank.conv(f"tail.0.{2 * st}.{j}", c.weight[j * F:(j + 1) * F], c.bias[j * F:(j + 1) * F], "c3")

    def _h16_ok(self):
        F = self.net.n_feats
        return F % 64 == 0 and F <= 1024 and all(e.use_planes and e.wp.fmt == 1 for k, e in self.bank.d.items() if e.kind == "c3")

    def forward_h16(self, x):
        net = self.net

        def attention(t, v, i):
            return t.enlca(v, [f"body.{i}.{s}" for s in ("conv_match1", "conv_match2", "conv_assembly")],
                           net.body[i].attn_fn.projection_matrix, net.res_scale)
        return self.edsr_body_h16(x, self._body(), attention)

    def graph(self, t, x3):
        net = self.net
        rs = net.res_scale
        x = t.conv_in1(x3, net.head[0].weight, net.head[0].bias, ("head.0.weight", "head.0.bias"))
        res = x
        for i, kind in self._body():
            m = net.body[i]
            if kind == "enlca":
                res = t.enlca(res, [f"body.{i}.{s}" for s in ("conv_match1", "conv_match2", "conv_assembly")],
                              m.attn_fn.projection_matrix, rs)
            elif kind == "res":
                r = t.conv(res, f"body.{i}.0", (f"body.{i}.body.0.weight", f"body.{i}.body.0.bias"), relu=True)
                res = t.conv(r, f"body.{i}.2", (f"body.{i}.body.2.weight", f"body.{i}.body.2.bias"), res=(res, rs))
            else:
                res = t.conv(res, f"body.{i}", (f"body.{i}.weight", f"body.{i}.bias"))
        res = t.axpby(res, x, 1.0, 1.0)
        for st in range(int(math.log2(net.upscale))):
            F = net.n_feats
            res = t.shuffle(t.cat([t.conv(res, f"tail.0.{2 * st}.{j}", (f"tail.0.{2 * st}.weight", f"tail.0.{2 * st}.bias",
                                                                       (j * F, (j + 1) * F))) for j in range(4)]), 2)
        return t.conv_out1(res, net.tail[1].weight, net.tail[1].bias, ("tail.1.weight", "tail.1.bias"))

