"""NLSN as a tape graph (reference dlib/models/network_nlsn.py:296-369): head conv; body = NonLocalSparseAttention, then
n_resblocks ResBlocks (conv-ReLU-conv, x res_scale, + x) with an attention block behind every eighth, a conv; long skip;
Upsampler (conv F -> 4F as four F-column convs + PixelShuffle(2) per factor of two) and the output conv.  Trains through the
tape's derived backward (the sparse attention: Tape.nlsa)."""
import math

from .tape import TapeEngine


def body_layout(n_resblocks):
    out, i = [(0, "nlsa")], 1
    for b in range(n_resblocks):
        out.append((i, "res")); i += 1
        if (b + 1) % 8 == 0:
            out.append((i, "nlsa")); i += 1
    out.append((i, "conv"))
    return out


class NLSNEngine(TapeEngine):
    rotations = None          # tests: one LSH rotation tensor per attention block (None: drawn per call, as the reference)
    taps = None               # tests: list that receives {"rotations", "order"} per attention block
    orders = None             # tests: one token order per attention block (replays another implementation's sort)

    def bank_entries(self):
        net, bank = self.net, self.bank
        F = net.n_feats
        for i, kind in body_layout(net.n_resblocks):
            m = net.body[i]
            if kind == "nlsa":
                bank.conv(f"body.{i}.conv_match", m.conv_match[0].weight, m.conv_match[0].bias, "c3")
                bank.conv(f"body.{i}.conv_assembly", m.conv_assembly[0].weight, m.conv_assembly[0].bias, "c1")
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
        state = {"a": 0}

        def attention(t, v, i):
            a = state["a"]
            state["a"] += 1
            tap = None
            if self.taps is not None:
                tap = {}
                self.taps.append(tap)
            return t.nlsa(v, (f"body.{i}.conv_match", f"body.{i}.conv_assembly"), net.n_hashes, net.chunk_size, net.res_scale,
                          None if self.rotations is None else self.rotations[a], tap,
                          None if self.orders is None else self.orders[a])
        layout = [(i, "nlsa" if k == "nlsa" else k) for i, k in body_layout(net.n_resblocks)]
        return self.edsr_body_h16(x, layout, attention)

    def graph(self, t, x3):
        net = self.net
        rs = net.res_scale
        x = t.conv_in1(x3, net.head[0].weight, net.head[0].bias, ("head.0.weight", "head.0.bias"))
        res, a = x, 0
        for i, kind in body_layout(net.n_resblocks):
            if kind == "nlsa":
                tap = None
                if self.taps is not None:
                    tap = {}
                    self.taps.append(tap)
                res = t.nlsa(res, (f"body.{i}.conv_match", f"body.{i}.conv_assembly"), net.n_hashes, net.chunk_size, rs,
                             None if self.rotations is None else self.rotations[a], tap,
                             None if self.orders is None else self.orders[a])
                a += 1
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

