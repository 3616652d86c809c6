"""MemNet on libsrhip (reference dlib/models/network_memnet.py:24-179; registry select_network.py:184-190):
same constructor, ``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)``, the reference's module tree -- hence its state_dict
keys, BatchNorm buffers included (``feature_extractor.{0,2}``, ``dense_memory_blocks.{i}.recursive_unit.{j}.
residual_block.{0,2,3,5}``, ``dense_memory_blocks.{i}.gate_unit.{0,2}``, ``reconstructor.{0,2}``) -- and its
initialisation (Kaiming-normal convs, BatchNorm weight 1); the modules only HOLD the parameters and buffers, the compute
is ``srhip.memnet_engine.MemNetEngine`` (BatchNorm: batch statistics + running-statistics update in train mode, running
statistics in eval mode).  1-channel inputs; GPU only (CPU tensors raise)."""
import torch
import torch.nn as nn

from srhip.module_path import refresh_if_params_changed

__all__ = ['MemNet']


class _ResidualBlock(nn.Module):                   # network_memnet.py:24-41
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.residual_block = nn.Sequential(
            nn.BatchNorm2d(channels), nn.ReLU(True), nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1), bias=False),
            nn.BatchNorm2d(channels), nn.ReLU(True), nn.Conv2d(channels, channels, (3, 3), (1, 1), (1, 1), bias=False))


class _MemoryBlock(nn.Module):                     # network_memnet.py:44-78
    def __init__(self, channels: int, num_memory_blocks: int, num_residual_blocks: int) -> None:
        super().__init__()
        gate_channels = int((num_residual_blocks + num_memory_blocks) * channels)
        self.num_residual_blocks = num_residual_blocks
        self.recursive_unit = nn.Sequential(*[_ResidualBlock(channels) for _ in range(num_residual_blocks)])
        self.gate_unit = nn.Sequential(nn.BatchNorm2d(gate_channels), nn.ReLU(True),
                                       nn.Conv2d(gate_channels, channels, (1, 1), (1, 1), (0, 0), bias=False))


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, need_grad, *params):
        ctx.net = net
        # --amp (inference only; training stays f32-accurate): the convs run ONE product of the operands' leading fp16 planes
        # (11 significant bits under the block exponents) -- PSNR within 0.003 dB of the f32-accurate forward on the
        # seeded-weights check (gate 0.01 dB, tests/test_gpu_amp.py; with one bf16 product, round 2, VDSR was 0.023 dB off)
        from srhip import ops
        with ops.amp_inference(getattr(net, "amp", False) and not need_grad):
            y = net.engine.forward(x, None, save=need_grad)
        return y.clone() if need_grad else y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        net.engine.backward(dy.contiguous(), grads)
        return (None, None, None) + tuple(grads[k] for k in names)


class MemNet(nn.Module):
    def __init__(self, in_chans: int, upscale: int, num_memory_blocks: int, num_residual_blocks: int):
        super().__init__()
        assert isinstance(upscale, int) and upscale > 0, upscale
        assert isinstance(in_chans, int) and in_chans > 0, in_chans
        if in_chans != 1:
            raise NotImplementedError("MemNet on libsrhip: 1-channel microscopy patches only")
        self.upscale, self.scale, self.in_chans = upscale, upscale, in_chans
        self.global_residual = None
        self.x_interp = None
        self.feature_extractor = nn.Sequential(nn.BatchNorm2d(in_chans), nn.ReLU(True),
                                               nn.Conv2d(in_chans, 64, (3, 3), (1, 1), (1, 1), bias=False))
        self.dense_memory_blocks = nn.Sequential(*[_MemoryBlock(64, i + 1, num_residual_blocks)
                                                   for i in range(num_memory_blocks)])
        self.reconstructor = nn.Sequential(nn.BatchNorm2d(64), nn.ReLU(True),
                                           nn.Conv2d(64, in_chans, (1, 1), (1, 1), (0, 0), bias=False))
        self._engine = None
        self._initialize_weights()

    def _initialize_weights(self) -> None:          # network_memnet.py:172-179
        for module in self.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight)
            elif isinstance(module, nn.BatchNorm2d):
                nn.init.constant_(module.weight, 1)

    def flush(self):
        self.global_residual = None
        self.x_interp = None

    @property
    def engine(self):
        if self._engine is None:
            from srhip.memnet_engine import MemNetEngine
            self._engine = MemNetEngine(self)
        return self._engine

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.invalidate()
        return out

    def weights_changed(self):
        if self._engine is not None:
            self._engine.invalidate()

    def sample_drop_path(self, batch, device):
        return None

    def prepare_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("MemNet (libsrhip) runs on the GPU only: move the model and the input to cuda; "
                               "there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, f'c: {x.shape}, img-nc: {self.in_chans}'
        return x.float().contiguous()[:, 0], x.shape[2], x.shape[3]

    def forward(self, x):
        self.flush()
        xi, h, w = self.prepare_input(x)
        params = [p for _, p in self.named_parameters()]
        # gradients exist in train mode (batch statistics); eval mode is inference
        need_grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in params)
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        if not self.training:
            # the running statistics are buffers: re-read them (80 tiny copies + one preparation launch) when something wrote
            # them since the last evaluation forward -- a training forward of the engine (it drops the flag itself: its
            # kernels write through raw pointers) or torch (load_state_dict, .copy_: the tensors' version counters).  Not in
            # every call: a host copy is not permitted inside a hipGraph capture (ModelPlain --eval_graph).
            sig = tuple(b._version for b in self.buffers())
            if sig != getattr(self, "_bn_buffer_versions", None):
                self._bn_buffer_versions = sig
                self.engine._eval_coefs = False
            gc_max = (len(self.dense_memory_blocks) + self.dense_memory_blocks[0].num_residual_blocks) * 64
            per = xi.shape[1] * xi.shape[2] * self.upscale ** 2 * gc_max
            # the widest gate concatenation of the batch must stay below 2^31 elements (32-bit staging offsets of the conv /
            # GEMM kernels); past that the batch is walked per image.  (6.4 GB at B = 8, 512 x 512: nothing on this part)
            if xi.shape[0] > 1 and xi.shape[0] * per >= (1 << 31):
                return torch.cat([_NetFn.apply(xi[b:b + 1], self, False, *params) for b in range(xi.shape[0])], 0)
        return _NetFn.apply(xi, self, need_grad, *params)
