"""Derived matmul operands of the evaluation-only engines written directly over the ops (GRL, ACT's Linears): a Linear / 1x1-conv
weight or the tap-major pack of a 3x3 conv, prepared once per weight version.  From 64 channels on the operand is split into
planes by the weight-preparation kernel (ops.PrepTable: two fp16 planes where the GEMM / conv kernels take them, three
bf16 planes otherwise) -- f32-grade with three / six products, ONE product under --amp (ops.amp_inference); narrower
operands stay exact f32."""
import torch

from . import ops


class PlaneCache:
    def __init__(self):
        self._w = {}

    def clear(self):
        self._w = {}

    def linear(self, key, w2d, bias):
        """(W operand for ops.gemm_nt, bias) of a weight [N, K]"""
        key = ("lin", key)
        if key not in self._w:
            w = w2d.contiguous()
            tb = None
            if ops.bx3_nt_for(*w.shape) and w.shape[1] % 4 == 0:
                P = ops.Bx3(w.shape[0], w.shape[1], w.device)
                tb = ops.PrepTable()
                tb.linear(w, P)
                tb.build(w.device).run()
                w = P
            self._w[key] = (w, bias, tb)
        return self._w[key][:2]

    def conv(self, key, w4d, bias, cin_pad=None, cout_pad=None):
        """(pack for ops.conv3x3, bias, Cout) of a 3x3 conv weight [Co, Ci, 3, 3], optionally zero-padded in either channel
        count (exact: the padded weights and biases are 0)"""
        key = ("conv", key)
        if key not in self._w:
            w, b = w4d, bias
            co, ci = w.shape[:2]
            cop, cip = cout_pad or co, cin_pad or ci
            if (cop, cip) != (co, ci):
                wz = torch.zeros(cop, cip, 3, 3, device=w.device)
                wz[:co, :ci] = w
                bz = torch.zeros(cop, device=w.device)
                if b is not None:
                    bz[:co] = b
                w, b = wz, bz
            w = w.contiguous()
            if ops.bx3_nt_for(cop, cip):
                wp = ops.Bx3(9 * cop, cip, w.device)
                tb = ops.PrepTable()
                tb.conv(w, wp)
                tb.build(w.device).run()
            else:
                wp, tb = torch.empty(9, cop, cip, device=w.device), None
                ops.pack_conv_weight(w, wp, None)
            self._w[key] = (wp, None if b is None else b.contiguous(), cop, tb)
        return self._w[key][:3]
