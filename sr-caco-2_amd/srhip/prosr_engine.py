"""ProSR as a tape graph (reference dlib/models/network_prosr.py:383-433): per pyramid level a chain of dense residual
blocks (dense layers = 1x1 conv, ReLU, reflection-padded 3x3 conv, concatenation; 1x1 compression; res_factor * block +
identity) with a level skip, a reflection-padded conv + PixelShuffle(2) + ReLU upsampler, and a reconstruction conv on
top of the bicubically interpolated input (stock F.interpolate, clamped, as the reference).  Reflection padding =
srhip_pad_reflect1, the zero-padded conv on the padded image, srhip_crop1."""
import torch
import torch.nn.functional as F

from .tape import TapeEngine


class ProSREngine(TapeEngine):
    def __init__(self, net):
        super().__init__(net)
        self.intermediate_outs = None

    def _levels(self):
        net = self.net
        for i in range(net.n_pyramids):
            yield i + 1, getattr(net, f"pyramid_residual_{i + 1}")

    def bank_entries(self):
        net, bank = self.net, self.bank
        for s, pyr in self._levels():
            pre = f"pyramid_residual_{s}"
            for name, mod in pyr.named_children():
                if name.startswith("compression_"):
                    bank.conv(f"{pre}.{name}", mod.conv1.weight, None, "c1")
                elif name.startswith("residual_denseblock_"):
                    for ln, layer in mod.dense_block.named_children():
                        bank.conv(f"{pre}.{name}.{ln}.c1", layer.conv_1.weight, layer.conv_1.bias, "c1")
                        c2 = layer.conv_2.conv[1]
                        bank.conv(f"{pre}.{name}.{ln}.c2", c2.weight, c2.bias, "c3")
                    bank.conv(f"{pre}.{name}.comp", mod.comp.conv1.weight, None, "c1")
                else:       # final_conv: Sequential([final_comp,] final_conv)
                    if hasattr(mod, "final_comp"):
                        bank.conv(f"{pre}.final_comp", mod.final_comp.conv1.weight, None, "c1")
                    c = mod.final_conv.conv[1]
                    bank.conv(f"{pre}.final", c.weight, c.bias, "c3")
            up = getattr(net, f"{pre}_residual_upsampler").m[0].conv[1]
            bank.conv(f"{pre}.up", up.weight, up.bias, "c3")

    def graph(self, t, x3):
        net = self.net
        B, H, W = x3.shape
        n = net.n_pyramids

        def rconv(v, key, pname):
            """reflection-padded 3x3 conv (network_prosr.py:38-86)"""
            return t.crop(t.conv(t.pad_reflect(v), key, (pname + ".weight", pname + ".bias")))

        # init conv of the requested (= largest) scale on the reflection-padded 1-channel image
        ic = getattr(net, f"init_conv_{n}").conv[1]
        xp = F.pad(x3[:, None], (1, 1, 1, 1), mode="reflect")[:, 0].contiguous()
        feats = t.crop(t.conv_in1(xp, ic.weight, ic.bias, (f"init_conv_{n}.conv.1.weight", f"init_conv_{n}.conv.1.bias")))
        outs = []
        for s, pyr in self._levels():
            pre = f"pyramid_residual_{s}"
            v = feats
            for name, mod in pyr.named_children():
                if name.startswith("compression_"):
                    v = t.conv(v, f"{pre}.{name}", (f"{pre}.{name}.conv1.weight", None))
                elif name.startswith("residual_denseblock_"):
                    ident = v
                    d = v
                    for ln, layer in mod.dense_block.named_children():
                        p = f"{pre}.{name}.dense_block.{ln}"
                        a = t.relu(t.conv(d, f"{pre}.{name}.{ln}.c1", (p + ".conv_1.weight", p + ".conv_1.bias")))
                        nf = rconv(a, f"{pre}.{name}.{ln}.c2", p + ".conv_2.conv.1")
                        d = t.cat([d, nf])
                    c = t.conv(d, f"{pre}.{name}.comp", (f"{pre}.{name}.comp.conv1.weight", None))
                    v = t.axpby(c, ident, net.res_factor, 1.0)
                else:
                    if hasattr(mod, "final_comp"):
                        v = t.conv(v, f"{pre}.final_comp", (f"{pre}.final_conv.final_comp.conv1.weight", None))
                    v = rconv(v, f"{pre}.final", f"{pre}.final_conv.final_conv.conv.1")
            feats = t.axpby(v, feats, 1.0, 1.0) if feats.t.shape == v.t.shape else v      # level skip (:393)
            u = t.shuffle(rconv(feats, f"{pre}.up", f"{pre}_residual_upsampler.m.0.conv.1"), 2)
            feats = u if net.ps_woReLU else t.relu(u)
            rc = getattr(net, f"reconst_{s}").final_conv.conv[1]
            z = t.crop1c(t.conv_out1(t.pad_reflect(feats), rc.weight, rc.bias,
                                     (f"reconst_{s}.final_conv.conv.1.weight", f"reconst_{s}.final_conv.conv.1.bias")))
            sc = 2 ** s
            ident = torch.clamp(F.interpolate(x3[:, None], size=(sc * H, sc * W), mode="bicubic", align_corners=False),
                                0.0, 1.0)[:, 0].contiguous()
            outs.append(t.add_const(z, ident))
        self._out_vars = outs
        self.intermediate_outs = [o.t.view(o.t.shape[0], 1, *o.t.shape[1:]) for o in outs[:-1]]
        return outs[-1]

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False, d_inter=None):
        if d_inter is not None:
            for v, g in zip(self._out_vars[:-1], d_inter):
                v.g = g.reshape(v.t.shape).contiguous()
        return super().backward(dy, grads, need_dx=need_dx)
