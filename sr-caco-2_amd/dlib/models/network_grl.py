"""GRL on libsrhip (reference dlib/models/network_grl.py:1113-1512; registry select_network.py:70-90): same constructor,
``forward((B,1,h,w) in [0,1]) -> (B,1,s*h,s*w)`` (inputs reflect-padded to multiples of the window, the output cropped) and
the reference's state_dict keys, shapes and order -- the 13 registered buffers (``table_*``, ``index_*``, ``mask_*``) first,
then ``conv_first``, ``norm_start``, ``layers.{i}.blocks.{j}.{attn.{qkv.body, anchor.body.0.reduction, window_attn.
attn_transform, stripe_attn.attn_transform{1,2}, proj}, norm1, conv.cab.{0,2,3.attention.{1,3}}, mlp.{fc1,fc2}, norm2}``,
``layers.{i}.conv``, ``norm_end``, ``conv_after_body``, ``conv_before_upsample.0``, ``upsample.up.{0,2,..}``, ``conv_last``.
Built for the options the registry passes: linear qkv / output projections, average-pooled anchors, '1conv' stage ends, the
pixel-shuffle upsampler, the local (conv + channel attention) branch on or off, no stripe shift.  The compute is
``srhip.grl_engine.GRLEngine``.  Training through the tape graph of the engine (window-multiple patches); 1-channel inputs; GPU only."""
import math

import torch
import torch.nn as nn

from dlib.models.network_dbpn import TapeNet

__all__ = ['GRL']


# ---------------------------------------------------------------- tables, indices, masks (network_grl.py:1534-1699)
def _partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws[0], ws[0], W // ws[1], ws[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws[0], ws[1], C)


def _coords(n):
    return torch.flatten(torch.stack(torch.meshgrid([torch.arange(0, n[0]), torch.arange(0, n[1])], indexing="ij")), 1)


def relative_position_index(ws, df=1, window_to_anchor=True):
    """get_relative_position_index_simple (:1553-1577)"""
    aws = [w // df for w in ws]
    c, ca = _coords(ws), _coords(aws)
    a, b, off = (c, ca, [w - 1 for w in aws]) if window_to_anchor else (ca, c, [w - 1 for w in ws])
    d = (a[:, :, None] - b[:, None, :]).permute(1, 2, 0).contiguous()
    d[:, :, 0] += off[0]
    d[:, :, 1] += off[1]
    d[:, :, 0] *= aws[1] + ws[1] - 1
    return d.sum(-1)


def relative_coords_table(ws, df=1):
    """get_relative_coords_table_all (:1651-1699) without a pretrained window size"""
    aws = [w // df for w in ws]
    hi = [w1 - 1 - (w1 - w2) // 2 for w1, w2 in zip(ws, aws)]
    lo = [-(w2 - 1) - (w1 - w2) // 2 for w1, w2 in zip(ws, aws)]
    t = torch.stack(torch.meshgrid([torch.arange(lo[0], hi[0] + 1, dtype=torch.float32),
                                    torch.arange(lo[1], hi[1] + 1, dtype=torch.float32)], indexing="ij"))
    t = t.permute(1, 2, 0).contiguous().unsqueeze(0)
    t[:, :, :, 0] /= hi[0]
    t[:, :, :, 1] /= hi[1]
    t *= 8
    return torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)


def _regions(res, ws, shift):
    m = torch.zeros((1, *res, 1))
    n = 0
    for h in (slice(0, -ws[0]), slice(-ws[0], -shift[0]), slice(-shift[0], None)):
        for w in (slice(0, -ws[1]), slice(-ws[1], -shift[1]), slice(-shift[1], None)):
            m[:, h, w, :] = n
            n += 1
    return _partition(m, ws).view(-1, ws[0] * ws[1])


def _as_mask(d):
    return d.masked_fill(d != 0, float(-100.0)).masked_fill(d == 0, float(0.0))


def shift_mask(res, ws, shift, df=1, window_to_anchor=None):
    """calculate_mask (:1607-1622; window_to_anchor None) / calculate_mask_all (:1625-1648)"""
    mw = _regions(res, ws, shift)
    if window_to_anchor is None:
        return _as_mask(mw.unsqueeze(1) - mw.unsqueeze(2))
    ma = _regions([s // df for s in res], [s // df for s in ws], [s // df for s in shift])
    return _as_mask(mw.unsqueeze(2) - ma.unsqueeze(1) if window_to_anchor else ma.unsqueeze(2) - mw.unsqueeze(1))


def table_index_mask(x_size, window_size, stripe_size, df):
    """GRL.set_table_index_mask (:1332-1375), in registration order"""
    ws, ss, rs = list(window_size), list(stripe_size), list(x_size)
    sv = ss[::-1]
    sh2, sv2 = [s // 2 for s in ss], [s // 2 for s in sv]
    return {
        "table_w": relative_coords_table(ws), "table_sh": relative_coords_table(ss, df), "table_sv": relative_coords_table(sv, df),
        "index_w": relative_position_index(ws),
        "index_sh_a2w": relative_position_index(ss, df, False), "index_sh_w2a": relative_position_index(ss, df, True),
        "index_sv_a2w": relative_position_index(sv, df, False), "index_sv_w2a": relative_position_index(sv, df, True),
        "mask_w": shift_mask(rs, ws, [w // 2 for w in ws]),
        "mask_sh_a2w": shift_mask(rs, ss, sh2, df, False), "mask_sh_w2a": shift_mask(rs, ss, sh2, df, True),
        "mask_sv_a2w": shift_mask(rs, sv, sv2, df, False), "mask_sv_w2a": shift_mask(rs, sv, sv2, df, True),
    }


# ---------------------------------------------------------------- parameter holders
class _Affine(nn.Module):                                         # AffineTransform :281-294, CPB_MLP :688-696
    def __init__(self, heads):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, heads, bias=False))


class _WindowAttn(nn.Module):
    def __init__(self, heads):
        super().__init__()
        self.attn_transform = _Affine(heads)


class _StripeAttn(nn.Module):
    def __init__(self, heads):
        super().__init__()
        self.attn_transform1 = _Affine(heads)
        self.attn_transform2 = _Affine(heads)


class _Body(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.body = m


class _AnchorLinear(nn.Module):                                   # :596-609
    def __init__(self, cin, cout):
        super().__init__()
        self.reduction = nn.Linear(cin, cout, bias=True)


class _MixedAttention(nn.Module):                                 # :822-859
    def __init__(self, dim, heads_w, heads_s, qkv_bias):
        super().__init__()
        self.qkv = _Body(nn.Linear(dim, dim * 3, bias=qkv_bias))
        self.anchor = _Body(nn.ModuleList([_AnchorLinear(dim, dim // 2)]))
        self.window_attn = _WindowAttn(heads_w)
        self.stripe_attn = _StripeAttn(heads_s)
        self.proj = nn.Linear(dim, dim)


class _ChannelAttention(nn.Module):                               # :713-726
    def __init__(self, c, reduction):
        super().__init__()
        self.attention = nn.Sequential(nn.Identity(), nn.Conv2d(c, c // reduction, 1), nn.ReLU(inplace=True),
                                       nn.Conv2d(c // reduction, c, 1), nn.Sigmoid())


class _CAB(nn.Module):                                            # :729-741
    def __init__(self, c, compress_ratio=4, reduction=18):
        super().__init__()
        self.cab = nn.Sequential(nn.Conv2d(c, c // compress_ratio, 3, 1, 1), nn.GELU(), nn.Conv2d(c // compress_ratio, c, 3, 1, 1),
                                 _ChannelAttention(c, reduction))


class _Mlp(nn.Module):
    def __init__(self, c, hidden):
        super().__init__()
        self.fc1 = nn.Linear(c, hidden)
        self.fc2 = nn.Linear(hidden, c)


class _Block(nn.Module):                                          # EfficientMixAttnTransformerBlock :940-1059
    def __init__(self, dim, heads_w, heads_s, mlp_ratio, qkv_bias, local_connection):
        super().__init__()
        self.attn = _MixedAttention(dim, heads_w, heads_s, qkv_bias)
        self.norm1 = nn.LayerNorm(dim)
        if local_connection:
            self.conv = _CAB(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.LayerNorm(dim)


class _Stage(nn.Module):                                          # TransformerStage :96-165
    def __init__(self, dim, depth, heads_w, heads_s, mlp_ratio, qkv_bias, local_connection):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(dim, heads_w, heads_s, mlp_ratio, qkv_bias, local_connection) for _ in range(depth)])
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1)


class _Upsample(nn.Module):                                       # :202-224
    def __init__(self, scale, nf):
        super().__init__()
        if scale & (scale - 1) or scale < 2:
            raise NotImplementedError(f"GRL on libsrhip: upscale {scale}: powers of two only")
        m = []
        for _ in range(int(math.log(scale, 2))):
            m += [nn.Conv2d(nf, 4 * nf, 3, 1, 1), nn.PixelShuffle(2)]
        self.up = nn.Sequential(*m)


class GRL(TapeNet):
    def __init__(self, img_size=64, in_chans=3, embed_dim=96, upscale=2, img_range=1.0, upsampler="", depths=[6, 6, 6, 6, 6, 6],
                 num_heads_window=[3, 3, 3, 3, 3, 3, 3], num_heads_stripe=[3, 3, 3, 3, 3, 3, 3], window_size=8, stripe_size=[8, 8],
                 stripe_groups=[None, None], stripe_shift=False, mlp_ratio=4.0, qkv_bias=True, qkv_proj_type="linear",
                 anchor_proj_type="avgpool", anchor_one_stage=True, anchor_window_down_factor=1, out_proj_type="linear",
                 local_connection=False, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 pretrained_window_size=[0, 0], pretrained_stripe_size=[0, 0], conv_type="1conv", init_method="n",
                 fairscale_checkpoint=False, offload_to_cpu=False, euclidean_dist=False, **kwargs):
        super().__init__()
        built = dict(upsampler="pixelshuffle", qkv_proj_type="linear", anchor_proj_type="avgpool", out_proj_type="linear",
                     conv_type="1conv", anchor_one_stage=True, stripe_shift=False, euclidean_dist=False, init_method="n",
                     fairscale_checkpoint=False)
        got = dict(upsampler=upsampler, qkv_proj_type=qkv_proj_type, anchor_proj_type=anchor_proj_type, out_proj_type=out_proj_type,
                   conv_type=conv_type, anchor_one_stage=anchor_one_stage, stripe_shift=stripe_shift, euclidean_dist=euclidean_dist,
                   init_method=init_method, fairscale_checkpoint=fairscale_checkpoint)
        for k, v in built.items():
            if got[k] != v:
                raise NotImplementedError(f"GRL on libsrhip: {k}={got[k]!r} is not built (the registry passes {v!r})")
        if list(stripe_groups) != [None, None] or list(pretrained_window_size) != [0, 0] or list(pretrained_stripe_size) != [0, 0]:
            raise NotImplementedError("GRL on libsrhip: stripe groups / pretrained window sizes are not built")
        if norm_layer is not nn.LayerNorm or float(img_range) != 1.0:
            raise NotImplementedError("GRL on libsrhip: nn.LayerNorm and img_range 1 only")
        self._init_protocol(upscale, in_chans)
        self.embed_dim, self.depths = embed_dim, list(depths)
        self.window_size = (window_size, window_size)
        self.stripe_size = list(stripe_size)
        self.df = anchor_window_down_factor
        self.heads_w, self.heads_s = list(num_heads_window), list(num_heads_stripe)
        self.local_connection = bool(local_connection)
        # stochastic depth (network_grl.py:1262, 1012, 1058-1066): rate i of linspace(0, drop_path_rate, sum(depths)) on both
        # residual branches of block i, timm's DropPath (per-sample Bernoulli(keep) / keep) in training mode
        self.drop_probs = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.pad_size = max(window_size, max(self.stripe_size))
        half = embed_dim // 2
        ok = (embed_dim % 4 == 0 and embed_dim >= 18 and window_size ** 2 <= 64 and self.stripe_size[0] * self.stripe_size[1] <= 64
              and all(s % self.df == 0 for s in self.stripe_size)
              and all(half % h == 0 and half // h <= 64 for h in self.heads_w[:len(depths)] + self.heads_s[:len(depths)]))
        if not ok:
            raise NotImplementedError("GRL on libsrhip: embed_dim a multiple of 4, heads dividing embed_dim / 2 into at most 64 "
                                      "channels, windows and stripes of at most 64 tokens")
        self.input_resolution = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        for k, v in table_index_mask(self.input_resolution, self.window_size, self.stripe_size, self.df).items():
            self.register_buffer(k, v)
        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.norm_start = nn.LayerNorm(embed_dim)
        self.layers = nn.ModuleList([_Stage(embed_dim, depths[i], num_heads_window[i], num_heads_stripe[i], mlp_ratio, qkv_bias,
                                            self.local_connection) for i in range(len(depths))])
        self.norm_end = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, 64, 3, 1, 1), nn.LeakyReLU(inplace=True))
        self.upsample = _Upsample(upscale, 64)
        self.conv_last = nn.Conv2d(64, in_chans, 3, 1, 1)
        for m in self.modules():                                  # GRL._init_weights :1400-1407
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def sample_drop_path(self, batch, device):
        """[2 nblocks, B] DropPath multipliers (one row per residual branch), None when nothing is dropped"""
        if not self.training or max(self.drop_probs) == 0.0:
            return None
        cache = getattr(self, "_dp_keep", None)
        if cache is None or cache.device != torch.device(device):
            cache = self._dp_keep = 1.0 - torch.tensor(self.drop_probs, device=device).repeat_interleave(2)
        mask = torch.bernoulli(cache[:, None].expand(-1, batch))
        return (mask / cache[:, None]).contiguous()

    def _make_engine(self):
        from srhip.grl_engine import GRLEngine
        return GRLEngine(self)
