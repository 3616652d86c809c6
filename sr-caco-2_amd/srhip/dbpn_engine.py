"""DBPN as a tape graph (reference dlib/models/network_dbpn.py:532-577): feat0 3x3 + PReLU, feat1 1x1 + PReLU, then
num_stages passes through the SAME seven up- and six down-projection units (weights shared between the passes), dense
concatenations of the HR / LR features inside a pass, and a 3x3 conv over the concatenated pass outputs."""
from .tape import TapeEngine


class DBPNEngine(TapeEngine):
    def _blocks(self):
        net = self.net
        out = [("up1", net.up1, True, False), ("down1", net.down1, False, False), ("up2", net.up2, True, False)]
        for i in range(2, 7):
            out.append((f"down{i}", getattr(net, f"down{i}"), False, True))
            out.append((f"up{i + 1}", getattr(net, f"up{i + 1}"), True, True))
        return out

    def bank_entries(self):
        net, bank = self.net, self.bank
        s, k, p = net.stride, net.kernel, net.padding
        bank.conv("feat1", net.feat1.conv.weight, net.feat1.conv.bias, "c1")
        for name, blk, up, dense in self._blocks():
            if dense:
                bank.conv(name + ".conv", blk.conv.conv.weight, blk.conv.conv.bias, "c1")
            a, b, c = (("up_conv1", "up_conv2", "up_conv3") if up else ("down_conv1", "down_conv2", "down_conv3"))
            for sub, is_deconv in ((a, up), (b, not up), (c, up)):
                m = getattr(blk, sub)
                if is_deconv:
                    bank.conv(f"{name}.{sub}", m.deconv.weight, m.deconv.bias, "deconv", s, k, p)
                else:
                    bank.conv(f"{name}.{sub}", m.conv.weight, m.conv.bias, "down", s, k, p)

    def _unit(self, t, name, blk, up, dense, x):
        """UpBlock / D_UpBlock: h0 = up(x); l0 = down(h0); h1 = up(l0 - x); h1 + h0   (DownBlock: the mirror image)."""
        if dense:
            x = t.conv(x, name + ".conv", (name + ".conv.conv.weight", name + ".conv.conv.bias"),
                       act=lambda v: t.prelu(v, blk.conv.act.weight, name + ".conv.act.weight"))
        subs = ("up_conv1", "up_conv2", "up_conv3") if up else ("down_conv1", "down_conv2", "down_conv3")

        def run(sub, v, is_deconv, add=None):
            m = getattr(blk, sub)
            inner = "deconv" if is_deconv else "conv"
            return t.conv(v, f"{name}.{sub}", (f"{name}.{sub}.{inner}.weight", f"{name}.{sub}.{inner}.bias"),
                          prelu=(m.act.weight, f"{name}.{sub}.act.weight"), add=add)
        a0 = run(subs[0], x, up)
        d0 = run(subs[1], a0, not up, add=(x, -1.0))         # l0 - x  (h0 - x)
        return run(subs[2], d0, up, add=(a0, 1.0))           # h1 + h0  (l1 + l0)

    def graph(self, t, x3):
        net = self.net
        f = t.conv_in1(x3, net.feat0.conv.weight, net.feat0.conv.bias, ("feat0.conv.weight", "feat0.conv.bias"))
        f = t.prelu(f, net.feat0.act.weight, "feat0.act.weight")
        l = t.conv(f, "feat1", ("feat1.conv.weight", "feat1.conv.bias"),
                   act=lambda v: t.prelu(v, net.feat1.act.weight, "feat1.act.weight"))
        B = dict((n, (b, up, dense)) for n, b, up, dense in self._blocks())

        def unit(name, x):
            b, up, dense = B[name]
            return self._unit(t, name, b, up, dense, x)
        results = []
        for _ in range(net.num_stages):                      # network_dbpn.py:537-570: the same modules every pass
            h1 = unit("up1", l)
            l1 = unit("down1", h1)
            h2 = unit("up2", l1)
            concat_h = t.cat([h2, h1])
            l = unit("down2", concat_h)
            concat_l = t.cat([l, l1])
            h = unit("up3", concat_l)
            for i in range(3, 7):
                concat_h = t.cat([h, concat_h])
                l = unit(f"down{i}", concat_h)
                concat_l = t.cat([l, concat_l])
                h = unit(f"up{i + 1}", concat_l)
            results.append(h)
        r = t.cat(results) if len(results) > 1 else results[0]
        return t.conv_out1(r, net.output_conv.conv.weight, net.output_conv.conv.bias,
                           ("output_conv.conv.weight", "output_conv.conv.bias"))
