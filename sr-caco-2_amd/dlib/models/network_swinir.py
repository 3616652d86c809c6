"""SwinIR on libsrhip.

Drop-in for the reference's ``dlib.models.network_swinir.SwinIR``
(network_swinir.py:710-970): same constructor arguments, same ``state_dict``
keys / shapes / order (366 entries for the README configuration), same
``forward(x: (B,C,h,w) in [0,1]) -> (B,C,h*s,w*s)`` contract.  The module only
*holds* parameters; ``forward`` runs the HIP engine (srhip/swinir_engine.py)
and there is no PyTorch/CPU execution path: CPU tensors raise.

Supported on the HIP path: window_size 8, 1-channel input, upsamplers
'pixelshuffledirect' (the configuration the reference trains and benchmarks,
README.md:120-197), 'nearest_conv' (x4: network_swinir.py:874-885,948-961) and 'pixelshuffle' (the registry default,
utils_init_default_args.py:23; conv 180->64 + LeakyReLU, log2(s) x [conv 64->256 +
PixelShuffle(2)], conv 64->1: network_swinir.py:862-868,937-942), resi_connection
'1conv' and '3conv' (network_swinir.py:543-552,849-858).
"""
import math

import torch
import torch.nn as nn

from srhip import ops
from srhip.module_path import refresh_if_params_changed
import torch.nn.functional as F

from dlib.utils import constants

__all__ = ['SwinIR']


class _Box(nn.Module):
    """Parameter container (gives the dotted state_dict names)."""


def _trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, mean=0., std=std, a=-2., b=2.)


def _linear(out_f, in_f, bias=True):
    m = _Box()
    m.weight = nn.Parameter(_trunc_normal_(torch.empty(out_f, in_f)))
    m.bias = nn.Parameter(torch.zeros(out_f)) if bias else None
    return m


def _norm(c):
    m = _Box()
    m.weight = nn.Parameter(torch.ones(c))
    m.bias = nn.Parameter(torch.zeros(c))
    return m


def _conv3(co, ci):
    m = _Box()
    bound = 1.0 / math.sqrt(ci * 9)   # nn.Conv2d default (kaiming_uniform, a=sqrt(5))
    m.weight = nn.Parameter((torch.rand(co, ci, 3, 3) * 2 - 1) * bound)
    m.bias = nn.Parameter((torch.rand(co) * 2 - 1) * bound)
    return m


def _conv1(co, ci):
    m = _Box()
    bound = 1.0 / math.sqrt(ci)       # nn.Conv2d(ci, co, 1) default
    m.weight = nn.Parameter((torch.rand(co, ci, 1, 1) * 2 - 1) * bound)
    m.bias = nn.Parameter((torch.rand(co) * 2 - 1) * bound)
    return m


def _resi_conv(dim, kind):
    """The conv in front of a residual connection (network_swinir.py:543-552, 849-858): '1conv' = Conv2d(dim, dim, 3);
    '3conv' = Sequential(Conv2d(dim, dim/4, 3), LeakyReLU(0.2), Conv2d(dim/4, dim/4, 1), LeakyReLU(0.2),
    Conv2d(dim/4, dim, 3)) -- parameter holders at the Sequential's indices 0, 2, 4."""
    if kind == constants.R_CONNECTION_1CONV:
        return _conv3(dim, dim)
    return nn.ModuleList([_conv3(dim // 4, dim), nn.Identity(), _conv1(dim // 4, dim // 4), nn.Identity(),
                          _conv3(dim, dim // 4)])


def _relative_position_index(ws):
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)


def _shift_mask(h, w, ws, shift):
    """{0,-100} mask buffer, kept only for state_dict compatibility
    (network_swinir.py:260-285); the kernel derives the mask from indices."""
    region = torch.zeros(h, w)
    k = 0
    for (h0, h1) in ((0, h - ws), (h - ws, h - shift), (h - shift, h)):
        for (w0, w1) in ((0, w - ws), (w - ws, w - shift), (w - shift, w)):
            region[h0:h1, w0:w1] = k
            k += 1
    r = region.reshape(h // ws, ws, w // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    d = r[:, None, :] - r[:, :, None]
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


class _SwinBlock(_Box):
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio,
                 drop_path, qkv_bias=True):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.drop_prob = float(drop_path)
        if self.shift_size > 0:
            self.register_buffer("attn_mask", _shift_mask(*input_resolution, window_size, shift_size))
        else:
            self.attn_mask = None
        self.norm1 = _norm(dim)
        attn = _Box()
        wsb = self.window_size
        attn.relative_position_bias_table = nn.Parameter(
            _trunc_normal_(torch.empty((2 * wsb - 1) ** 2, num_heads)))
        attn.register_buffer("relative_position_index", _relative_position_index(wsb))
        attn.qkv = _linear(3 * dim, dim, bias=qkv_bias)
        attn.proj = _linear(dim, dim)
        self.attn = attn
        self.norm2 = _norm(dim)
        mlp = _Box()
        mlp.fc1 = _linear(int(dim * mlp_ratio), dim)
        mlp.fc2 = _linear(dim, int(dim * mlp_ratio))
        self.mlp = mlp


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network (autograd-compatible path;
    the training loop in ModelPlain drives the engine directly)."""

    @staticmethod
    def forward(ctx, x, net, dp, need_grad, *params):
        ctx.net = net
        ctx.need_dx = x.requires_grad
        # reduced-precision matmuls only for inference (net.amp, set by --amp): training stays f32-accurate
        with ops.amp_inference(getattr(net, "amp", False) and not need_grad):
            y = net.engine.forward(x, dp, save=need_grad)
        return y.clone() if need_grad else y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        names = [k for k, _ in net.named_parameters()]
        grads = {k: torch.empty_like(p) for k, p in net.named_parameters()}
        dx = net.engine.backward(dy.contiguous(), grads, need_dx=ctx.need_dx)
        return (dx, None, None, None) + tuple(grads[k] for k in names)


class SwinIR(nn.Module):
    # forward() multiplies the input by img_range and divides the output by it (reference :935,968): the fused training
    # step (srhip/train.py) applies the same scale around the loss only for a net that says so
    forward_divides_by_img_range = True

    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=[6, 6, 6, 6],
                 num_heads=[6, 6, 6, 6], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, use_checkpoint=False, upscale=2, img_range=1.,
                 upsampler='', resi_connection=constants.R_CONNECTION_1CONV, **kwargs):
        super().__init__()
        unsupported = []
        # window_size != 8 or a given qk_scale: the general tape graph (srhip/swinir_tape_engine.py); otherwise the fused engine
        self.use_tape = window_size != 8 or qk_scale is not None
        self.qk_scale = qk_scale
        if not 1 <= in_chans <= 4:
            unsupported.append(f"in_chans={in_chans} (HIP path: 1 to 4 image channels)")
        if upsampler not in (constants.US_PIXEL_SHUFFLE_DIRECT, constants.US_PIXEL_SHUFFLE, constants.US_NEAREST_CONV):
            unsupported.append(f"upsampler={upsampler!r} (HIP path: 'pixelshuffledirect', 'pixelshuffle', 'nearest_conv')")
        if upsampler == constants.US_NEAREST_CONV:
            assert upscale == 4, 'only support x4 now.'              # network_swinir.py:876
        if upsampler == constants.US_PIXEL_SHUFFLE and (upscale & (upscale - 1) or upscale < 2):
            unsupported.append(f"upscale={upscale} with 'pixelshuffle' (HIP path: powers of two)")
        if resi_connection not in (constants.R_CONNECTION_1CONV, constants.R_CONNECTION_3CONV):
            unsupported.append(f"resi_connection={resi_connection!r} (HIP path: '1conv', '3conv')")
        if not self.use_tape and (embed_dim > 256 or (embed_dim // num_heads[0]) not in (10, 16, 30, 32)):
            unsupported.append(f"embed_dim={embed_dim} / heads={num_heads}")
        if self.use_tape and (embed_dim % 4 or any(embed_dim % h for h in num_heads)):
            unsupported.append(f"embed_dim={embed_dim} / heads={num_heads} (a multiple of 4, divisible by the heads)")
        if unsupported:
            raise NotImplementedError("SwinIR on libsrhip does not support: " + "; ".join(unsupported))
        size = img_size if isinstance(img_size, (tuple, list)) else (img_size, img_size)
        self.img_size = tuple(size)
        self.in_chans, self.embed_dim, self.mlp_ratio = in_chans, embed_dim, mlp_ratio
        self.depths, self.num_heads = list(depths), list(num_heads)
        self.window_size, self.upscale, self.img_range = window_size, upscale, img_range
        self.upsampler = upsampler
        self.resi_connection = resi_connection
        if in_chans == 3:       # network_swinir.py:722-727
            self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
        else:
            self.mean = torch.zeros(1, 1, 1, 1)
        # nn.Dropout (pos_drop, Mlp.drop, proj_drop, attn_drop; network_swinir.py:48-56, 113-117, 817) is the identity in
        # evaluation mode, which runs as is; a training-mode forward with a non-zero rate raises (sample_drop_path)
        self.drop_rate, self.attn_drop_rate = float(drop_rate), float(attn_drop_rate)
        self.ape = bool(ape)
        if self.ape:    # network_swinir.py:812-815 (patch_size 1: one row per pixel of an img_size patch)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.img_size[0] * self.img_size[1], embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)

        self.conv_first = _conv3(embed_dim, in_chans)
        self.patch_embed = _Box()
        self.patch_norm = bool(patch_norm)
        if self.patch_norm:                     # network_swinir.py:799-803: no norm -> no parameters in patch_embed
            self.patch_embed.norm = _norm(embed_dim)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for li, depth in enumerate(depths):
            rstb = _Box()
            rstb.residual_group = _Box()
            rstb.residual_group.blocks = nn.ModuleList([
                _SwinBlock(embed_dim, self.img_size, num_heads[li], window_size,
                           0 if j % 2 == 0 else window_size // 2, mlp_ratio,
                           dpr[sum(depths[:li]) + j], qkv_bias=qkv_bias) for j in range(depth)])
            rstb.conv = _resi_conv(embed_dim, resi_connection)
            self.layers.append(rstb)
        self.norm = _norm(embed_dim)
        self.conv_after_body = _resi_conv(embed_dim, resi_connection)
        if upsampler == constants.US_PIXEL_SHUFFLE_DIRECT:
            self.upsample = nn.ModuleList([_conv3(upscale * upscale * in_chans, embed_dim)])
        elif upsampler == constants.US_NEAREST_CONV:   # parameter names / order of network_swinir.py:874-885
            num_feat = 64
            self.num_feat = num_feat
            self.conv_before_upsample = nn.ModuleList([_conv3(num_feat, embed_dim)])
            self.conv_up1 = _conv3(num_feat, num_feat)
            self.conv_up2 = _conv3(num_feat, num_feat)
            self.conv_hr = _conv3(num_feat, num_feat)
            self.conv_last = _conv3(in_chans, num_feat)
        else:       # 'pixelshuffle': same parameter names / order as network_swinir.py:862-868
            num_feat = 64
            self.num_feat = num_feat
            self.conv_before_upsample = nn.ModuleList([_conv3(num_feat, embed_dim)])
            up = []
            for _ in range(int(math.log2(upscale))):
                up += [_conv3(4 * num_feat, num_feat), nn.Identity()]     # upsample.{0,2,4}: conv; odd: PixelShuffle
            self.upsample = nn.ModuleList(up)
            self.conv_last = _conv3(in_chans, num_feat)
        self._engine = None

    # -- helpers -------------------------------------------------------------
    def swin_blocks(self):
        for layer in self.layers:
            for b in layer.residual_group.blocks:
                yield b

    @property
    def engine(self):
        if self._engine is None:
            if self.use_tape:
                from srhip.swinir_tape_engine import SwinIRTapeEngine
                self._engine = SwinIRTapeEngine(self)
            else:
                from srhip.swinir_engine import SwinIREngine
                self._engine = SwinIREngine(self)
        return self._engine

    def _apply(self, fn, *a, **k):   # .cuda() / .to(): parameter storage moves
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.invalidate()
        return out

    def weights_changed(self):
        """Call after parameters were modified in place (optimizer step)."""
        if self._engine is not None:
            self._engine.invalidate()

    def sample_drop_path(self, batch, device):
        """timm DropPath semantics (per-sample Bernoulli(keep)/keep), one row per
        (block, branch); None when nothing is dropped."""
        if self.training and (self.drop_rate or self.attn_drop_rate):
            raise NotImplementedError("SwinIR on libsrhip: drop_rate / attn_drop_rate > 0 in training mode (the dropout masks "
                                      "are not implemented; evaluation, where dropout is the identity, runs)")
        probs = [b.drop_prob for b in self.swin_blocks()]
        if not self.training or max(probs) == 0.0:
            return None
        # the keep probabilities live on the device (a per-step host->device copy would
        # synchronise the stream and stop the CPU from running ahead of the GPU)
        cache = getattr(self, "_dp_keep", None)
        if cache is None or cache[0] != probs or cache[1].device != torch.device(device):
            keep = 1.0 - torch.tensor(probs, device=device).repeat_interleave(2)
            cache = self._dp_keep = (probs, keep)
        keep = cache[1]
        mask = torch.bernoulli(keep[:, None].expand(-1, batch))
        return (mask / keep[:, None]).contiguous()

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    # -- forward -------------------------------------------------------------
    def prepare_input(self, x):
        """check_image_size (network_swinir.py:908-913): reflect-pad to a multiple of the window; (x - mean) * img_range
        (:934-935; the mean is zero unless in_chans == 3).  in_chans 1: [B, H, W]; otherwise NHWC with the channels
        zero-padded to 4 (what the engine's conv_first takes)."""
        if not x.is_cuda:
            raise RuntimeError("SwinIR (libsrhip) runs on the GPU only: move the model and the "
                               "input to cuda; there is no CPU fallback")
        assert x.dim() == 4 and x.shape[1] == self.in_chans, x.shape
        _, _, h, w = x.shape
        ws = self.window_size
        ph, pw = (ws - h % ws) % ws, (ws - w % ws) % ws
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph), 'reflect')
        if self.in_chans == 3:
            x = x - self.mean.to(x)
        if self.img_range != 1.:
            x = x * self.img_range
        if self.in_chans == 1:
            return x.float().contiguous()[:, 0], h, w
        return F.pad(x.float().permute(0, 2, 3, 1), (0, 4 - self.in_chans)).contiguous(), h, w

    def forward(self, x, dp=None):
        xi, h, w = self.prepare_input(x)
        if dp is None:
            dp = self.sample_drop_path(x.shape[0], x.device)
        if xi.shape[1] <= self.window_size or xi.shape[2] <= self.window_size:
            raise NotImplementedError("inputs must be larger than one window")
        if self.ape and tuple(xi.shape[1:3]) != self.img_size:   # the reference's broadcast fails the same way (:919)
            raise RuntimeError(f"ape=True: the input has to be img_size {self.img_size}, got {tuple(xi.shape[1:3])}")
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and (xi.requires_grad or any(p.requires_grad for p in params))
        refresh_if_params_changed(self, params)   # stock torch.optim wrote the weights?
        y = _NetFn.apply(xi, self, dp, need_grad, *params)
        if self.img_range != 1.:
            y = y / self.img_range
        if self.in_chans == 3:
            y = y + self.mean.to(y)
        s = self.upscale
        return y[:, :, :h * s, :w * s]
