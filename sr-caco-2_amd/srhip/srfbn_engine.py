"""SRFBN as a tape graph (reference dlib/models/network_srfbn.py:475-586,651-679): conv_in 3x3 + PReLU, feat_in 1x1 +
PReLU, then num_steps passes through ONE feedback block (its output is the next pass's hidden state: the gradient
runs back through the passes), every pass reconstructed by a transposed conv + 3x3 conv on top of the bilinearly
interpolated input (stock F.interpolate, as the reference)."""
import torch
import torch.nn.functional as F

from .tape import TapeEngine


class SRFBNEngine(TapeEngine):
    def __init__(self, net):
        super().__init__(net)
        self.all_outs = []
        self.intermediate_outs = None       # TrainStep: the trainer's curriculum loss averages every pass's prediction

    def bank_entries(self):
        net, bank = self.net, self.bank
        s, k, p = net.stride, net.kernel, net.padding
        bank.conv("feat_in", net.feat_in[0].weight, net.feat_in[0].bias, "c1")
        blk = net.block
        bank.conv("cin", blk.compress_in[0].weight, blk.compress_in[0].bias, "c1")
        bank.conv("cout", blk.compress_out[0].weight, blk.compress_out[0].bias, "c1")
        for i in range(blk.num_groups):
            bank.conv(f"up{i}", blk.upBlocks[i][0].weight, blk.upBlocks[i][0].bias, "deconv", s, k, p)
            bank.conv(f"down{i}", blk.downBlocks[i][0].weight, blk.downBlocks[i][0].bias, "down", s, k, p)
            if i > 0:
                bank.conv(f"upt{i}", blk.uptranBlocks[i - 1][0].weight, blk.uptranBlocks[i - 1][0].bias, "c1")
                bank.conv(f"downt{i}", blk.downtranBlocks[i - 1][0].weight, blk.downtranBlocks[i - 1][0].bias, "c1")
        bank.conv("out", net.out[0].weight, net.out[0].bias, "deconv", s, k, p)

    def graph(self, t, x3):
        net, blk = self.net, self.net.block
        s = net.upscale
        inter = F.interpolate(x3[:, None], scale_factor=s, mode='bilinear', align_corners=False)[:, 0].contiguous()

        def cna(v, key, pre):
            """conv + PReLU of Sequential `pre` (parameters pre.0.weight / pre.0.bias / pre.1.weight)."""
            mod = net.get_submodule(pre)
            return t.conv(v, key, (pre + ".0.weight", pre + ".0.bias"), prelu=(mod[1].weight, pre + ".1.weight"))
        x = t.conv_in1(x3, net.conv_in[0].weight, net.conv_in[0].bias, ("conv_in.0.weight", "conv_in.0.bias"))
        x = t.prelu(x, net.conv_in[1].weight, "conv_in.1.weight")
        x = cna(x, "feat_in", "feat_in")
        hidden = x                                            # should_reset: last_hidden = x (network_srfbn.py:541-544)
        outs = []
        for _ in range(net.num_steps):
            c = cna(t.cat([x, hidden]), "cin", "block.compress_in")
            lr, hr = [c], []
            for i in range(blk.num_groups):
                L = t.cat(lr) if len(lr) > 1 else lr[0]
                if i > 0:
                    L = cna(L, f"upt{i}", f"block.uptranBlocks.{i - 1}")
                Hh = cna(L, f"up{i}", f"block.upBlocks.{i}")
                hr.append(Hh)
                Hc = t.cat(hr) if len(hr) > 1 else hr[0]
                if i > 0:
                    Hc = cna(Hc, f"downt{i}", f"block.downtranBlocks.{i - 1}")
                lr.append(cna(Hc, f"down{i}", f"block.downBlocks.{i}"))
            o = t.cat(lr[1:]) if len(lr) > 2 else lr[1]
            hidden = cna(o, "cout", "block.compress_out")
            h = cna(hidden, "out", "out")
            y = t.conv_out1(h, net.conv_out[0].weight, net.conv_out[0].bias, ("conv_out.0.weight", "conv_out.0.bias"))
            outs.append(t.add_const(y, inter))
        self._out_vars = outs
        B, H, W = outs[-1].t.shape
        self.all_outs = [o.t.view(B, 1, H, W) for o in outs]
        self.intermediate_outs = self.all_outs[:-1]
        return outs[-1]

    def backward(self, dy, grads, need_dx=False, on_layer_done=None, grads_zeroed=False, d_inter=None):
        if d_inter is not None:                               # gradients of the earlier passes' predictions
            for v, g in zip(self._out_vars[:-1], d_inter):
                v.g = g.reshape(v.t.shape).contiguous()
        return super().backward(dy, grads, need_dx=need_dx)
