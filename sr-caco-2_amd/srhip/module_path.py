"""Helpers of the nn.Module ("drop-in") path of the network mirrors.

The engines keep DERIVED operands (LayerNorm-folded weights, bf16x3 planes, conv
packs, bias images) that are rebuilt by prepare() only when someone says the
parameters changed.  TrainStep / ModelPlain say so explicitly; a caller that
trains the module with a stock ``torch.optim`` optimizer does not -- so the
module path looks for itself before every forward: parameter storage moved
(``data_ptr``) or parameters written in place (``_version``, bumped by every
in-place aten op, optimizer.step() included)."""


def param_signature(params):
    return tuple((p.data_ptr(), p._version) for p in params)


def refresh_if_params_changed(net, params):
    """Invalidate net's engine when the parameters differ from the ones the derived
    operands were built from.  ~0.1 ms for SwinIR's 330 tensors; module path only."""
    sig = param_signature(params)
    if getattr(net, "_param_sig", None) != sig:
        net._param_sig = sig
        net.weights_changed()
