"""DFCAN as a tape graph (reference dlib/models/network_dfcan.py:86-116): conv + GELU; four residual groups of four RCABs
(two conv + GELU, the Fourier channel attention, skip); conv 64 -> 64 s^2 + GELU as 256-column slices; PixelShuffle(s);
conv + sigmoid.  Trains through the tape's derived backward (Tape.fourier_gate: the spectrum magnitude through the in-tree DFT passes, dfca.hip)."""
from .tape import TapeEngine


class DFCANEngine(TapeEngine):
    def bank_entries(self):
        net, bank = self.net, self.bank
        for g in range(4):
            for r in range(4):
                m = net.RGs[g].RCABs[r]
                pre = f"RGs.{g}.RCABs.{r}"
                for sub in ("conv_gelu1", "conv_gelu2", "conv_relu1"):
                    c = getattr(m, sub)[0]
                    bank.conv(f"{pre}.{sub}", c.weight, c.bias, "c3")
        c = net.conv_gelu[0]
        co = c.weight.shape[0]
        self.nslices = max(1, co // 256)
        w = co // self.nslices
        for j in range(self.nslices):
            bank.conv(f"conv_gelu.{j}", c.weight[j * w:(j + 1) * w], c.bias[j * w:(j + 1) * w], "c3")

    def graph(self, t, x3):
        net = self.net
        x = t.conv_in1(x3, net.input[0].weight, net.input[0].bias, ("input.0.weight", "input.0.bias"))
        x = t.unary(x, "gelu")
        for g in range(4):
            x0 = x
            for r in range(4):
                m = net.RGs[g].RCABs[r]
                pre = f"RGs.{g}.RCABs.{r}"
                a = t.conv(x, f"{pre}.conv_gelu1", (f"{pre}.conv_gelu1.0.weight", f"{pre}.conv_gelu1.0.bias"), gelu=True)
                b = t.conv(a, f"{pre}.conv_gelu2", (f"{pre}.conv_gelu2.0.weight", f"{pre}.conv_gelu2.0.bias"), gelu=True)
                w1, w2 = m.conv_relu2[0].weight, m.conv_sigmoid[0].weight
                x = t.fourier_gate(x, b, f"{pre}.conv_relu1", (f"{pre}.conv_relu1.0.weight", f"{pre}.conv_relu1.0.bias"),
                                   w1.data.reshape(w1.shape[0], w1.shape[1]).contiguous(), m.conv_relu2[0].bias.data,
                                   w2.data.reshape(w2.shape[0], w2.shape[1]).contiguous(), m.conv_sigmoid[0].bias.data,
                                   names_gate=(f"{pre}.conv_relu2.0.weight", f"{pre}.conv_relu2.0.bias",
                                               f"{pre}.conv_sigmoid.0.weight", f"{pre}.conv_sigmoid.0.bias"))
            x = t.axpby(x, x0, 1.0, 1.0)
        wsl = net.conv_gelu[0].weight.shape[0] // self.nslices
        parts = [t.conv(x, f"conv_gelu.{j}", ("conv_gelu.0.weight", "conv_gelu.0.bias", (j * wsl, (j + 1) * wsl)))
                 for j in range(self.nslices)]
        u = t.unary(t.cat(parts) if len(parts) > 1 else parts[0], "gelu")
        u = t.shuffle(u, net.upscale)
        y = t.conv_out1(u, net.conv_sigmoid[0].weight, net.conv_sigmoid[0].bias, ("conv_sigmoid.0.weight", "conv_sigmoid.0.bias"))
        return t.unary(y, "sigmoid")

