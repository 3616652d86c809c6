"""CPU oracle for the SR-CACO-2 patch-level hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-fp32 (metrics: fp64) restatement of the reference
algorithms on the hot path.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``sr-caco-2_amd/`` imports it; the product path
runs hand-written HIP kernels and fails loudly when they are missing.

Parity pin: ``oracle/make_goldens.py`` imports the real reference modules from
``/root/reference`` (through a ``sys.modules`` shim), checks every function
below against them on seeded inputs, and stores the reference outputs as small
fixtures in ``tests/golden/``.  ``tests/test_oracle_golden.py`` re-checks the
oracle against those fixtures wherever the reference tree is absent.

Everything here is *functional*: networks take a ``state_dict`` (same keys and
shapes as the reference modules) and a config dict.  Citations are
``file:line`` relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# index-only pieces (bit-exact gates)
# ----------------------------------------------------------------------------
def pixel_shuffle(x: Tensor, r: int) -> Tensor:
    """out[b,c,h*r+i,w*r+j] = in[b,c*r*r+i*r+j,h,w]  (nn.PixelShuffle as used at
    dlib/models/network_nlsn.py:108 and network_swinir.py:701)."""
    b, c, h, w = x.shape
    co = c // (r * r)
    x = x.reshape(b, co, r, r, h, w)
    return x.permute(0, 1, 4, 2, 5, 3).reshape(b, co, h * r, w * r)


def window_partition(x: Tensor, ws: int) -> Tensor:
    """(B,H,W,C) -> (B*nW, ws, ws, C); network_swinir.py:48-62."""
    b, h, w, c = x.shape
    x = x.reshape(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(win: Tensor, ws: int, h: int, w: int) -> Tensor:
    """(B*nW, ws, ws, C) -> (B,H,W,C); network_swinir.py:65-80."""
    b = win.shape[0] // ((h // ws) * (w // ws))
    x = win.reshape(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def relative_position_index(ws: int) -> Tensor:
    """(ws*ws, ws*ws) int64 table index; network_swinir.py:116-128."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shifted_window_mask(h: int, w: int, ws: int, shift: int) -> Tensor:
    """(nW, ws*ws, ws*ws) in {0,-100}; network_swinir.py:260-285."""
    region = torch.zeros(h, w)
    bounds_h = [(0, h - ws), (h - ws, h - shift), (h - shift, h)]
    bounds_w = [(0, w - ws), (w - ws, w - shift), (w - shift, w)]
    k = 0
    for (h0, h1) in bounds_h:
        for (w0, w1) in bounds_w:
            region[h0:h1, w0:w1] = k
            k += 1
    rw = window_partition(region[None, :, :, None], ws).reshape(-1, ws * ws)
    diff = rw[:, None, :] - rw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0),
                       torch.zeros_like(diff))


# ----------------------------------------------------------------------------
# SwinIR (network_swinir.py:710-970)
# ----------------------------------------------------------------------------
def swinir_config(upscale=8, in_chans=1, img_size=64, window_size=8,
                  img_range=1.0, depths=(6, 6, 6, 6), embed_dim=180,
                  num_heads=(6, 6, 6, 6), mlp_ratio=2,
                  upsampler="pixelshuffledirect", resi_connection="1conv",
                  drop_path_rate=0.1, ape=False, patch_norm=True, qkv_bias=True, qk_scale=None) -> dict:
    """README.md:120-197 configuration by default."""
    return dict(ape=ape, patch_norm=patch_norm, qkv_bias=qkv_bias, qk_scale=qk_scale, upscale=upscale, in_chans=in_chans, img_size=img_size,
                window_size=window_size, img_range=img_range,
                depths=list(depths), embed_dim=embed_dim,
                num_heads=list(num_heads), mlp_ratio=mlp_ratio,
                upsampler=upsampler, resi_connection=resi_connection,
                drop_path_rate=drop_path_rate)


def swinir_drop_path_rates(cfg: dict) -> List[float]:
    """network_swinir.py:821 -- linspace(0, rate, sum(depths))."""
    n = sum(cfg["depths"])
    return [v.item() for v in torch.linspace(0, cfg["drop_path_rate"], n)]


def _wmsa(sd: SD, pre: str, xw: Tensor, heads: int, mask: Optional[Tensor],
          rpi: Tensor, taps: Optional[dict], qk_scale: Optional[float] = None) -> Tensor:
    """WindowAttention.forward, network_swinir.py:140-179."""
    nb, n, c = xw.shape
    d = c // heads
    qkv = F.linear(xw, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))   # qkv_bias=False: no such key (:104)
    if taps is not None:
        taps.setdefault("qkv", qkv.detach().clone())
    qkv = qkv.reshape(nb, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (qk_scale or d ** -0.5), qkv[1], qkv[2]        # :102 (scale = qk_scale or head_dim ** -0.5)
    att = q @ k.transpose(-2, -1)
    bias = sd[pre + "relative_position_bias_table"][rpi.reshape(-1)]
    att = att + bias.reshape(n, n, heads).permute(2, 0, 1)[None]
    if mask is not None:
        nw = mask.shape[0]
        att = (att.reshape(nb // nw, nw, heads, n, n)
               + mask[None, :, None]).reshape(nb, heads, n, n)
    att = att.softmax(dim=-1)
    if taps is not None:
        taps.setdefault("attn_probs_w0", att[0].detach().clone())
    out = (att @ v).transpose(1, 2).reshape(nb, n, c)
    return F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def _swin_block(sd: SD, pre: str, x: Tensor, hw: Tuple[int, int], ws: int,
                shift: int, heads: int, dp_scale: Optional[Tensor],
                rpi: Tensor, taps: Optional[dict], qk_scale: Optional[float] = None) -> Tensor:
    """SwinTransformerBlock.forward, network_swinir.py:287-337."""
    h, w = hw
    b, l, c = x.shape
    y = F.layer_norm(x, (c,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    if taps is not None:
        taps.setdefault("ln1", y.detach().clone())
    y = y.reshape(b, h, w, c)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
        mask = shifted_window_mask(h, w, ws, shift).to(x)
    else:
        mask = None
    yw = window_partition(y, ws).reshape(-1, ws * ws, c)
    yw = _wmsa(sd, pre + "attn.", yw, heads, mask, rpi, taps, qk_scale)
    y = window_reverse(yw.reshape(-1, ws, ws, c), ws, h, w)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y.reshape(b, l, c)
    if dp_scale is not None:  # timm DropPath: per-sample mask / keep_prob
        y = y * dp_scale[0].reshape(b, 1, 1)
    x = x + y
    y = F.layer_norm(x, (c,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    y = F.linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    y = F.gelu(y)  # nn.GELU default = exact erf (network_swinir.py:30)
    y = F.linear(y, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if dp_scale is not None:
        y = y * dp_scale[1].reshape(b, 1, 1)
    x = x + y
    if taps is not None:
        taps.setdefault("block0_out", x.detach().clone())
    return x


def _resi_conv(sd: SD, name: str, img: Tensor, cfg: dict) -> Tensor:
    """The conv in front of a residual connection (network_swinir.py:543-552, 849-858): '1conv' = one 3x3 conv;
    '3conv' = conv3x3 C -> C/4, LeakyReLU(0.2), conv1x1, LeakyReLU(0.2), conv3x3 C/4 -> C."""
    if cfg.get("resi_connection", "1conv") == "1conv":
        return F.conv2d(img, sd[name + ".weight"], sd[name + ".bias"], padding=1)
    v = F.leaky_relu(F.conv2d(img, sd[name + ".0.weight"], sd[name + ".0.bias"], padding=1), 0.2)
    v = F.leaky_relu(F.conv2d(v, sd[name + ".2.weight"], sd[name + ".2.bias"]), 0.2)
    return F.conv2d(v, sd[name + ".4.weight"], sd[name + ".4.bias"], padding=1)


def swinir_forward(sd: SD, x: Tensor, cfg: dict,
                   dp_scales: Optional[Sequence[Tensor]] = None,
                   taps: Optional[dict] = None) -> Tensor:
    """SwinIR.forward for upsampler 'pixelshuffledirect' / 'pixelshuffle' / 'nearest+conv'
    (network_swinir.py:930-970).  ``dp_scales``: per block a (2,B) tensor of
    DropPath multipliers (mask/keep_prob) for the attention and MLP branches,
    or None for eval / drop_path_rate 0.  ``taps`` collects intermediates of
    the first block for per-stage parity tests."""
    ws = cfg["window_size"]
    c = cfg["embed_dim"]
    s = cfg["upscale"]
    h0, w0 = x.shape[2:]
    ph, pw = (ws - h0 % ws) % ws, (ws - w0 % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph), mode="reflect")  # :908-913
    mean = torch.zeros(1, 1, 1, 1) if cfg["in_chans"] != 3 else \
        torch.tensor((0.4488, 0.4371, 0.4040)).reshape(1, 3, 1, 1)
    mean = mean.to(x)
    x = (x - mean) * cfg["img_range"]
    h, w = x.shape[2:]
    rpi = relative_position_index(ws)

    f0 = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    t = f0.flatten(2).transpose(1, 2)  # PatchEmbed :610-614
    if "patch_embed.norm.weight" in sd:      # patch_norm=True (:611-612)
        t = F.layer_norm(t, (c,), sd["patch_embed.norm.weight"],
                         sd["patch_embed.norm.bias"])
    if cfg.get("ape", False):    # network_swinir.py:918-919 (the input has to be img_size x img_size)
        t = t + sd["absolute_pos_embed"]
    bi = 0
    for li, depth in enumerate(cfg["depths"]):
        t_in = t
        for j in range(depth):
            # a block clamps its window to the image when the image is not
            # larger than the window (network_swinir.py:232-236)
            wsj, shift = ws, (0 if j % 2 == 0 else ws // 2)
            if cfg["img_size"] <= ws:
                wsj, shift = cfg["img_size"], 0
            t = _swin_block(sd, f"layers.{li}.residual_group.blocks.{j}.", t,
                            (h, w), wsj, shift, cfg["num_heads"][li],
                            None if dp_scales is None else dp_scales[bi],
                            rpi if wsj == ws else relative_position_index(wsj),
                            taps if bi == 0 else None, cfg.get("qk_scale"))
            bi += 1
        img = t.transpose(1, 2).reshape(-1, c, h, w)  # PatchUnEmbed :651-655
        img = _resi_conv(sd, f"layers.{li}.conv", img, cfg)
        t = img.flatten(2).transpose(1, 2) + t_in  # RSTB :562-565
    t = F.layer_norm(t, (c,), sd["norm.weight"], sd["norm.bias"])
    img = t.transpose(1, 2).reshape(-1, c, h, w)
    f = _resi_conv(sd, "conv_after_body", img, cfg) + f0
    if cfg["upsampler"] == "pixelshuffledirect":  # :943-947
        y = F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"],
                     padding=1)
        y = pixel_shuffle(y, s)
    elif cfg["upsampler"] == "pixelshuffle":  # :937-942
        y = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"],
                                  sd["conv_before_upsample.0.bias"], padding=1),
                         0.01)
        for i in range(int(math.log2(s))):
            y = F.conv2d(y, sd[f"upsample.{2 * i}.weight"],
                         sd[f"upsample.{2 * i}.bias"], padding=1)
            y = pixel_shuffle(y, 2)
        y = F.conv2d(y, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    elif cfg["upsampler"] == "nearest_conv":  # :948-961 (x4 only, :876)
        lr = lambda v: F.leaky_relu(v, 0.2)
        y = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"],
                                  sd["conv_before_upsample.0.bias"], padding=1), 0.01)
        y = lr(F.conv2d(F.interpolate(y, scale_factor=2, mode="nearest"), sd["conv_up1.weight"],
                        sd["conv_up1.bias"], padding=1))
        y = lr(F.conv2d(F.interpolate(y, scale_factor=2, mode="nearest"), sd["conv_up2.weight"],
                        sd["conv_up2.bias"], padding=1))
        y = F.conv2d(lr(F.conv2d(y, sd["conv_hr.weight"], sd["conv_hr.bias"], padding=1)),
                     sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    else:
        raise NotImplementedError(cfg["upsampler"])
    y = y / cfg["img_range"] + mean
    return y[:, :, :h0 * s, :w0 * s]


def swinir_init_state_dict(cfg: dict, seed: int = 0) -> SD:
    """Deterministic weights with the reference's shapes/keys.  Values follow
    the reference's init *distributions* (trunc-normal std .02 for Linear and
    the bias table, default Conv2d init; network_swinir.py:891-897) but not its
    RNG stream -- parity tests always load one state_dict into both sides."""
    g = torch.Generator().manual_seed(seed)
    c, ws = cfg["embed_dim"], cfg["window_size"]
    hid = int(c * cfg["mlp_ratio"])
    sd: SD = {}

    def conv(name, co, ci, k=3):
        bound = 1.0 / math.sqrt(ci * k * k)
        sd[name + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def resi(name):              # network_swinir.py:543-552
        if cfg.get("resi_connection", "1conv") == "1conv":
            conv(name, c, c)
        else:
            conv(name + ".0", c // 4, c)
            conv(name + ".2", c // 4, c // 4, 1)
            conv(name + ".4", c, c // 4)

    def lin(name, co, ci):
        sd[name + ".weight"] = torch.nn.init.trunc_normal_(
            torch.empty(co, ci), std=0.02, generator=g)
        sd[name + ".bias"] = torch.zeros(co)

    def ln(name):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)

    size = cfg["img_size"]
    if cfg.get("ape", False):    # network_swinir.py:812-815; a module's own parameters precede its children's
        sd["absolute_pos_embed"] = torch.nn.init.trunc_normal_(
            torch.empty(1, size * size, c), std=0.02, generator=g)
    conv("conv_first", c, cfg["in_chans"])
    if cfg.get("patch_norm", True):
        ln("patch_embed.norm")
    wsb = min(ws, size) if size <= ws else ws
    for li, depth in enumerate(cfg["depths"]):
        heads = cfg["num_heads"][li]
        for j in range(depth):
            p = f"layers.{li}.residual_group.blocks.{j}."
            shift = 0 if (j % 2 == 0 or size <= ws) else ws // 2
            if shift:  # a module's own buffers precede its children
                sd[p + "attn_mask"] = shifted_window_mask(size, size, ws, shift)
            ln(p + "norm1")
            sd[p + "attn.relative_position_bias_table"] = \
                torch.nn.init.trunc_normal_(
                    torch.empty((2 * wsb - 1) ** 2, heads), std=0.02, generator=g)
            sd[p + "attn.relative_position_index"] = relative_position_index(wsb)
            lin(p + "attn.qkv", 3 * c, c)
            if not cfg.get("qkv_bias", True):
                del sd[p + "attn.qkv.bias"]
            lin(p + "attn.proj", c, c)
            ln(p + "norm2")
            lin(p + "mlp.fc1", hid, c)
            lin(p + "mlp.fc2", c, hid)
        resi(f"layers.{li}.conv")
    ln("norm")
    resi("conv_after_body")
    s = cfg["upscale"]
    if cfg["upsampler"] == "pixelshuffledirect":
        conv("upsample.0", s * s * cfg["in_chans"], c)
    elif cfg["upsampler"] == "nearest_conv":     # network_swinir.py:874-885
        conv("conv_before_upsample.0", 64, c)
        for name in ("conv_up1", "conv_up2", "conv_hr"):
            conv(name, 64, 64)
        conv("conv_last", cfg["in_chans"], 64)
    else:
        conv("conv_before_upsample.0", 64, c)
        for i in range(int(math.log2(s))):
            conv(f"upsample.{2 * i}", 256, 64)
        conv("conv_last", cfg["in_chans"], 64)
    return sd


def trained_like_(sd: SD, seed: int, lin_scale: float = 10.0, table_std: float = 1.0,
                  bias_std: float = 0.1, conv_scale: float = 1.0, qk_scale: float = 1.0) -> SD:
    """In place: move a freshly initialised state_dict into a trained-like regime (test
    infrastructure; no reference counterpart -- the reference ships no weights here).  Fresh
    weights (trunc-normal 0.02, zero biases) leave the net nearly linear and every softmax
    nearly uniform, so the {0,-100} shift mask, large logits and the GELU tails are never
    exercised.  Linear weights x lin_scale, relative-position tables ~ N(0, table_std),
    Linear biases and LayerNorm beta ~ N(0, bias_std), LayerNorm gamma ~ 1 + N(0, bias_std),
    conv weights x conv_scale."""
    g = torch.Generator().manual_seed(seed)
    for k, v in sd.items():
        if v.dtype != torch.float32 or k.endswith("attn_mask"):
            continue
        if k.endswith("relative_position_bias_table"):
            v.copy_(torch.randn(v.shape, generator=g) * table_std)
        elif "norm" in k:
            v.add_(bias_std * torch.randn(v.shape, generator=g))
        elif v.ndim == 2:
            v.mul_(lin_scale)
            if k.endswith("qkv.weight") and qk_scale != 1.0:   # q and k rows only: sharper logits
                v[:2 * v.shape[0] // 3].mul_(qk_scale)
        elif v.ndim == 4:
            v.mul_(conv_scale)
        elif v.ndim == 1 and ("qkv" in k or "proj" in k or "fc" in k):
            v.add_(bias_std * torch.randn(v.shape, generator=g))
    return sd


def srcnn_forward(sd: SD, x: Tensor) -> Tensor:
    """SRCNN._forward_impl (network_srcnn.py:54-60): conv 5x5 (1 -> 1024) + ReLU, conv 1x1 (-> 128) + ReLU,
    conv 1x1 (-> 1), on an input already at the target size."""
    out = F.relu(F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], padding=2))
    out = F.relu(F.conv2d(out, sd["map.0.weight"], sd["map.0.bias"]))
    return F.conv2d(out, sd["reconstruction.weight"], sd["reconstruction.bias"])


def srcnn_init_state_dict(in_chans: int = 1, seed: int = 0, bias_std: float = 0.0) -> SD:
    """network_srcnn.py:63-72 as distributions: N(0, sqrt(2 / (out_channels * k*k))), the reconstruction layer
    N(0, 0.001), zero biases (bias_std > 0 perturbs them for the tests)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for name, co, ci, k in (("features.0", 1024, in_chans, 5), ("map.0", 128, 1024, 1), ("reconstruction", in_chans, 128, 1)):
        std = 0.001 if name == "reconstruction" else math.sqrt(2 / (co * k * k))
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * std
        sd[name + ".bias"] = torch.randn(co, generator=g) * bias_std
    return sd


# ----------------------------------------------------------------------------
# EDSR-baseline assembled from the reference's EDSR blocks
# (network_nlsn.py:38-128 blocks, :355-369 wiring without attention;
#  sizes utils_init_default_args.py:37-50)
# ----------------------------------------------------------------------------
def edsr_config(upscale=4, in_chans=1, n_feats=64, n_resblocks=16,
                res_scale=1.0) -> dict:
    return dict(upscale=upscale, in_chans=in_chans, n_feats=n_feats,
                n_resblocks=n_resblocks, res_scale=res_scale)


def edsr_forward(sd: SD, x: Tensor, cfg: dict, relu_masks=None) -> Tensor:
    """relu_masks (tests only): per ResBlock a {0, 1} tensor that REPLACES the ReLU's own decision -- the ReLU becomes
    pre * mask.  A gradient check against another computation of the same net is then not confused by pixels whose
    pre-activation lies within rounding of zero (one flipped decision moves a weight-gradient entry by a whole pixel's
    contribution): both sides differentiate the same piecewise-linear function."""
    nb = cfg["n_resblocks"]
    f0 = F.conv2d(x, sd["head.0.weight"], sd["head.0.bias"], padding=1)
    r = f0
    for k in range(nb):  # ResBlock: conv-ReLU-conv, *res_scale, +x  (:89-93)
        y = F.conv2d(r, sd[f"body.{k}.body.0.weight"], sd[f"body.{k}.body.0.bias"], padding=1)
        y = F.relu(y) if relu_masks is None else y * relu_masks[k].to(y.dtype)
        y = F.conv2d(y, sd[f"body.{k}.body.2.weight"],
                     sd[f"body.{k}.body.2.bias"], padding=1)
        r = y * cfg["res_scale"] + r
    r = F.conv2d(r, sd[f"body.{nb}.weight"], sd[f"body.{nb}.bias"], padding=1)
    r = r + f0  # :363-364
    for i in range(int(math.log2(cfg["upscale"]))):  # Upsampler :100-108
        r = F.conv2d(r, sd[f"tail.0.{2 * i}.weight"],
                     sd[f"tail.0.{2 * i}.bias"], padding=1)
        r = pixel_shuffle(r, 2)
    return F.conv2d(r, sd["tail.1.weight"], sd["tail.1.bias"], padding=1)


def edsr_init_state_dict(cfg: dict, seed: int = 0) -> SD:
    g = torch.Generator().manual_seed(seed)
    nf, nb = cfg["n_feats"], cfg["n_resblocks"]
    sd: SD = {}

    def conv(name, co, ci):
        bound = 1.0 / math.sqrt(ci * 9)
        sd[name + ".weight"] = (torch.rand(co, ci, 3, 3, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    conv("head.0", nf, cfg["in_chans"])
    for k in range(nb):
        conv(f"body.{k}.body.0", nf, nf)
        conv(f"body.{k}.body.2", nf, nf)
    conv(f"body.{nb}", nf, nf)
    for i in range(int(math.log2(cfg["upscale"]))):
        conv(f"tail.0.{2 * i}", 4 * nf, nf)
    conv("tail.1", cfg["in_chans"], nf)
    return sd


# ----------------------------------------------------------------------------
# VDSR (dlib/models/network_vdsr.py)
# ----------------------------------------------------------------------------
def vdsr_forward(sd: SD, x: Tensor, upscale: int) -> Tensor:
    """network_vdsr.py:78-117: bicubic up (align_corners False, clamped), conv1 + ReLU, 18 x (conv + ReLU),
    conv2, + the interpolated input.  No biases."""
    xi = torch.clamp(F.interpolate(x, size=(upscale * x.shape[2], upscale * x.shape[3]), mode="bicubic",
                                   align_corners=False), 0.0, 1.0)
    out = F.relu(F.conv2d(xi, sd["conv1.0.weight"], padding=1))
    k = 0
    while f"trunk.{k}.conv.weight" in sd:
        out = F.relu(F.conv2d(out, sd[f"trunk.{k}.conv.weight"], padding=1))
        k += 1
    return F.conv2d(out, sd["conv2.weight"], padding=1) + xi


def vdsr_init_state_dict(in_chans: int = 1, seed: int = 0) -> SD:
    """network_vdsr.py:121-126: N(0, sqrt(2 / (9 * Cout))) for every conv."""
    g = torch.Generator().manual_seed(seed)
    sd = {"conv1.0.weight": torch.randn(64, in_chans, 3, 3, generator=g) * math.sqrt(2 / (9 * 64))}
    for k in range(18):
        sd[f"trunk.{k}.conv.weight"] = torch.randn(64, 64, 3, 3, generator=g) * math.sqrt(2 / (9 * 64))
    sd["conv2.weight"] = torch.randn(in_chans, 64, 3, 3, generator=g) * math.sqrt(2 / (9 * in_chans))
    return sd


# ----------------------------------------------------------------------------
# DBPN (dlib/models/network_dbpn.py)
# ----------------------------------------------------------------------------
_DBPN_KSP = {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}


def _dbpn_conv(sd: SD, pre: str, x: Tensor, stride: int, padding: int, act: bool = True) -> Tensor:
    """ConvBlock (network_dbpn.py:71-105, norm None): conv (+ PReLU)."""
    y = F.conv2d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"], stride=stride, padding=padding)
    return F.prelu(y, sd[pre + ".act.weight"]) if act else y


def _dbpn_deconv(sd: SD, pre: str, x: Tensor, stride: int, padding: int) -> Tensor:
    """DeconvBlock (:108-143): ConvTranspose2d + PReLU."""
    y = F.conv_transpose2d(x, sd[pre + ".deconv.weight"], sd[pre + ".deconv.bias"], stride=stride, padding=padding)
    return F.prelu(y, sd[pre + ".act.weight"])


def _dbpn_up(sd: SD, pre: str, x: Tensor, s: int, p: int, dense: bool) -> Tensor:
    """UpBlock / D_UpBlock (:190-205, :225-246): h0 = up(x); l0 = down(h0); h1 = up(l0 - x); h1 + h0."""
    if dense:
        x = _dbpn_conv(sd, pre + ".conv", x, 1, 0)
    h0 = _dbpn_deconv(sd, pre + ".up_conv1", x, s, p)
    l0 = _dbpn_conv(sd, pre + ".up_conv2", h0, s, p)
    h1 = _dbpn_deconv(sd, pre + ".up_conv3", l0 - x, s, p)
    return h1 + h0


def _dbpn_down(sd: SD, pre: str, x: Tensor, s: int, p: int, dense: bool) -> Tensor:
    """DownBlock / D_DownBlock (:272-287, :306-327): l0 = down(x); h0 = up(l0); l1 = down(h0 - x); l1 + l0."""
    if dense:
        x = _dbpn_conv(sd, pre + ".conv", x, 1, 0)
    l0 = _dbpn_conv(sd, pre + ".down_conv1", x, s, p)
    h0 = _dbpn_deconv(sd, pre + ".down_conv2", l0, s, p)
    l1 = _dbpn_conv(sd, pre + ".down_conv3", h0 - x, s, p)
    return l1 + l0


def dbpn_forward(sd: SD, x: Tensor, upscale: int, num_stages: int = 3) -> Tensor:
    """DBPN.forward (network_dbpn.py:532-577): the seven up / six down units are applied num_stages times with the
    SAME weights, the pass outputs are concatenated and reconstructed by a 3x3 conv."""
    _, s, p = _DBPN_KSP[upscale]
    x = _dbpn_conv(sd, "feat0", x, 1, 1)
    l = _dbpn_conv(sd, "feat1", x, 1, 0)
    results = []
    for _ in range(num_stages):
        h1 = _dbpn_up(sd, "up1", l, s, p, False)
        l1 = _dbpn_down(sd, "down1", h1, s, p, False)
        h2 = _dbpn_up(sd, "up2", l1, s, p, False)
        concat_h = torch.cat((h2, h1), 1)
        l = _dbpn_down(sd, "down2", concat_h, s, p, True)
        concat_l = torch.cat((l, l1), 1)
        h = _dbpn_up(sd, "up3", concat_l, s, p, True)
        for i in range(3, 7):
            concat_h = torch.cat((h, concat_h), 1)
            l = _dbpn_down(sd, f"down{i}", concat_h, s, p, True)
            concat_l = torch.cat((l, concat_l), 1)
            h = _dbpn_up(sd, f"up{i + 1}", concat_l, s, p, True)
        results.append(h)
    return _dbpn_conv(sd, "output_conv", torch.cat(results, 1), 1, 1, act=False)


def dbpn_init_state_dict(upscale: int, in_chans: int = 1, base_filter: int = 64, feat: int = 256, num_stages: int = 3,
                         seed: int = 0, bias_std: float = 0.0) -> SD:
    """network_dbpn.py:445-530: kaiming-normal conv / deconv weights (std sqrt(2 / fan_in), fan_in = dim 1 x k x k --
    for a ConvTranspose2d weight [Cin, Cout, k, k] torch's fan_in is Cout k k), zero biases, PReLU slopes 0.25.
    bias_std / a jitter on the slopes (tests) make every parameter's gradient path visible."""
    k, s, p = _DBPN_KSP[upscale]
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(pre, ci, co, kk, act=True):
        sd[pre + ".conv.weight"] = torch.randn(co, ci, kk, kk, generator=g) * math.sqrt(2.0 / (ci * kk * kk))
        sd[pre + ".conv.bias"] = torch.randn(co, generator=g) * bias_std
        if act:
            sd[pre + ".act.weight"] = torch.full((1,), 0.25) + (torch.rand(1, generator=g) - 0.5) * (0.2 if bias_std else 0.0)

    def deconv(pre, ci, co, kk):
        sd[pre + ".deconv.weight"] = torch.randn(ci, co, kk, kk, generator=g) * math.sqrt(2.0 / (co * kk * kk))
        sd[pre + ".deconv.bias"] = torch.randn(co, generator=g) * bias_std
        sd[pre + ".act.weight"] = torch.full((1,), 0.25) + (torch.rand(1, generator=g) - 0.5) * (0.2 if bias_std else 0.0)

    def up(pre, dense):
        if dense:
            conv(pre + ".conv", base_filter * dense, base_filter, 1)
        deconv(pre + ".up_conv1", base_filter, base_filter, k)
        conv(pre + ".up_conv2", base_filter, base_filter, k)
        deconv(pre + ".up_conv3", base_filter, base_filter, k)

    def down(pre, dense):
        if dense:
            conv(pre + ".conv", base_filter * dense, base_filter, 1)
        conv(pre + ".down_conv1", base_filter, base_filter, k)
        deconv(pre + ".down_conv2", base_filter, base_filter, k)
        conv(pre + ".down_conv3", base_filter, base_filter, k)

    conv("feat0", in_chans, feat, 3)
    conv("feat1", feat, base_filter, 1)
    up("up1", 0)
    down("down1", 0)
    up("up2", 0)
    for i in range(2, 7):
        down(f"down{i}", i)
        up(f"up{i + 1}", i)
    conv("output_conv", num_stages * base_filter, in_chans, 3, act=False)
    return sd


# ----------------------------------------------------------------------------
# SRFBN (dlib/models/network_srfbn.py)
# ----------------------------------------------------------------------------
_SRFBN_KSP = {2: (6, 2, 2), 3: (7, 3, 2), 4: (8, 4, 2), 8: (12, 8, 2)}


def srfbn_forward(sd: SD, x: Tensor, upscale: int, num_steps: int = 4, num_groups: int = 6) -> List[Tensor]:
    """SRFBN.forward (network_srfbn.py:651-679) with FeedbackBlock.forward (:540-576): returns the predictions of all
    num_steps passes (the last one is the network output; model_plain.py:202-232 averages the loss over all of them)."""
    _, s, p = _SRFBN_KSP[upscale]

    def cna(pre, v, stride=1, padding=0, transposed=False):
        if transposed:
            y = F.conv_transpose2d(v, sd[pre + ".0.weight"], sd[pre + ".0.bias"], stride=stride, padding=padding)
        else:
            y = F.conv2d(v, sd[pre + ".0.weight"], sd[pre + ".0.bias"], stride=stride, padding=padding)
        return F.prelu(y, sd[pre + ".1.weight"])
    inter = F.interpolate(x, scale_factor=upscale, mode="bilinear", align_corners=False)
    x = cna("conv_in", x, 1, 1)
    x = cna("feat_in", x)
    hidden = x
    outs = []
    for _ in range(num_steps):
        c = cna("block.compress_in", torch.cat((x, hidden), 1))
        lr, hr = [c], []
        for i in range(num_groups):
            L = torch.cat(tuple(lr), 1)
            if i > 0:
                L = cna(f"block.uptranBlocks.{i - 1}", L)
            Hh = cna(f"block.upBlocks.{i}", L, s, p, transposed=True)
            hr.append(Hh)
            Hc = torch.cat(tuple(hr), 1)
            if i > 0:
                Hc = cna(f"block.downtranBlocks.{i - 1}", Hc)
            lr.append(cna(f"block.downBlocks.{i}", Hc, s, p))
        hidden = cna("block.compress_out", torch.cat(tuple(lr[1:]), 1))
        h = cna("out", hidden, s, p, transposed=True)
        outs.append(inter + F.conv2d(h, sd["conv_out.0.weight"], sd["conv_out.0.bias"], padding=1))
    return outs


def srfbn_init_state_dict(upscale: int, in_chans: int = 1, num_features: int = 64, num_groups: int = 6,
                          seed: int = 0) -> SD:
    """Seeded weights of the reference's layout (the reference keeps torch's default initialisation; fixtures load a
    state_dict, never compare fresh initialisations): N(0, 1 / sqrt(fan_in)) weights, small biases, PReLU slopes near 0.2;
    the frozen MeanShift convs as network_srfbn.py:123-133."""
    k, _, _ = _SRFBN_KSP[upscale]
    nf = num_features
    g = torch.Generator().manual_seed(seed)
    sd: SD = {"sub_mean.weight": torch.eye(3).view(3, 3, 1, 1), "sub_mean.bias": -255. * torch.tensor([0.4488, 0.4371, 0.4040])}

    def blk(pre, ci, co, kk, transposed=False, act=True):
        shape = (ci, co, kk, kk) if transposed else (co, ci, kk, kk)
        fan = (co if transposed else ci) * kk * kk
        sd[pre + ".0.weight"] = torch.randn(*shape, generator=g) / math.sqrt(fan)
        sd[pre + ".0.bias"] = torch.randn(co, generator=g) * 0.05
        if act:
            sd[pre + ".1.weight"] = torch.full((1,), 0.2) + (torch.rand(1, generator=g) - 0.5) * 0.2

    blk("conv_in", in_chans, 4 * nf, 3)
    blk("feat_in", 4 * nf, nf, 1)
    blk("block.compress_in", 2 * nf, nf, 1)
    for i in range(num_groups):
        blk(f"block.upBlocks.{i}", nf, nf, k, transposed=True)
    for i in range(num_groups):
        blk(f"block.downBlocks.{i}", nf, nf, k)
    for i in range(1, num_groups):
        blk(f"block.uptranBlocks.{i - 1}", nf * (i + 1), nf, 1)
    for i in range(1, num_groups):
        blk(f"block.downtranBlocks.{i - 1}", nf * (i + 1), nf, 1)
    blk("block.compress_out", num_groups * nf, nf, 1)
    blk("out", nf, nf, k, transposed=True)
    blk("conv_out", nf, in_chans, 3, act=False)
    sd["add_mean.weight"] = torch.eye(3).view(3, 3, 1, 1)
    sd["add_mean.bias"] = 255. * torch.tensor([0.4488, 0.4371, 0.4040])
    return sd


# ----------------------------------------------------------------------------
# ProSR (dlib/models/network_prosr.py), residual-dense-block form
# ----------------------------------------------------------------------------
def prosr_config(upscale=8, in_chans=1, num_init_features=160, bn_size=4, growth_rate=40, ps_woReLU=False,
                 level_config=None, level_compression=-1, res_factor=0.2, max_num_feature=312) -> dict:
    if level_config is None:        # utils_init_default_args.py: hard-coded per scale
        level_config = {2: [[8] * 9], 4: [[8] * 9, [8] * 3], 8: [[8] * 9, [8] * 3, [8]]}[upscale]
    return dict(upscale=upscale, in_chans=in_chans, num_init_features=num_init_features, bn_size=bn_size,
                growth_rate=growth_rate, ps_woReLU=ps_woReLU, level_config=level_config,
                level_compression=level_compression, res_factor=res_factor, max_num_feature=max_num_feature)


def _rconv(sd: SD, pre: str, x: Tensor) -> Tensor:
    """Conv2d of network_prosr.py:38-86: ReflectionPad2d(1) + 3x3 conv."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[pre + ".conv.1.weight"], sd[pre + ".conv.1.bias"])


def prosr_forward(sd: SD, x: Tensor, cfg: dict) -> List[Tensor]:
    """ProSR.forward (network_prosr.py:383-433) at the largest scale, no blending: the predictions of every pyramid level
    (the last one is the output; the others are intermediate_outs)."""
    n = int(math.log2(cfg["upscale"]))
    feats = _rconv(sd, f"init_conv_{n}", x)
    nf = cfg["num_init_features"]
    outs = []
    for i in range(n):
        pre = f"pyramid_residual_{i + 1}"
        v = feats
        if i != 0:
            v = F.conv2d(v, sd[f"{pre}.compression_{i}.conv1.weight"])
            nf = v.shape[1]
        for b, nl in enumerate(cfg["level_config"][i]):
            bp = f"{pre}.residual_denseblock_{b + 1}"
            d = v
            for l in range(nl):
                lp = f"{bp}.dense_block.denselayer{l + 1}"
                a = F.relu(F.conv2d(d, sd[lp + ".conv_1.weight"], sd[lp + ".conv_1.bias"]))
                d = torch.cat([d, _rconv(sd, lp + ".conv_2", a)], 1)
            v = cfg["res_factor"] * F.conv2d(d, sd[bp + ".comp.conv1.weight"]) + v
        if nf > cfg["max_num_feature"]:
            v = F.conv2d(v, sd[f"{pre}.final_conv.final_comp.conv1.weight"])
            nf = cfg["max_num_feature"]
        v = _rconv(sd, f"{pre}.final_conv.final_conv", v)
        feats = v + feats
        feats = F.pixel_shuffle(_rconv(sd, f"{pre}_residual_upsampler.m.0", feats), 2)
        if not cfg["ps_woReLU"]:
            feats = F.relu(feats)
        z = _rconv(sd, f"reconst_{i + 1}.final_conv", feats)
        sc = 2 ** (i + 1)
        ident = torch.clamp(F.interpolate(x, size=(sc * x.shape[2], sc * x.shape[3]), mode="bicubic",
                                          align_corners=False), 0.0, 1.0)
        outs.append(z + ident)
    return outs


def prosr_init_state_dict(cfg: dict, seed: int = 0, bias_std: float = 0.0) -> SD:
    """Seeded weights in the reference's state_dict layout and ORDER (N(0, 1 / sqrt(fan_in)); biases zero as init_weights
    leaves them, or small for tests)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(pre, ci, co, k, bias=True):
        sd[pre + ".weight"] = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
        if bias:
            sd[pre + ".bias"] = torch.randn(co, generator=g) * bias_std

    n = int(math.log2(cfg["upscale"]))
    nf0 = cfg["num_init_features"]
    for s in range(1, n + 1):
        conv(f"init_conv_{s}.conv.1", cfg["in_chans"], nf0, 3)
    nf = nf0
    gr, bs = cfg["growth_rate"], cfg["bn_size"]
    for i in range(n):
        pre = f"pyramid_residual_{i + 1}"
        if i != 0:
            out_planes = nf0 if cfg["level_compression"] <= 0 else int(cfg["level_compression"] * nf)
            conv(f"{pre}.compression_{i}.conv1", nf, out_planes, 1, bias=False)
            nf = out_planes
        for b, nl in enumerate(cfg["level_config"][i]):
            bp = f"{pre}.residual_denseblock_{b + 1}"
            for l in range(nl):
                lp = f"{bp}.dense_block.denselayer{l + 1}"
                conv(lp + ".conv_1", nf + l * gr, bs * gr, 1)
                conv(lp + ".conv_2.conv.1", bs * gr, gr, 3)
            conv(bp + ".comp.conv1", nf + nl * gr, nf, 1, bias=False)
        if nf > cfg["max_num_feature"]:
            conv(f"{pre}.final_conv.final_comp.conv1", nf, cfg["max_num_feature"], 1, bias=False)
            nf = cfg["max_num_feature"]
        conv(f"{pre}.final_conv.final_conv.conv.1", nf, nf, 3)
        conv(f"{pre}_residual_upsampler.m.0.conv.1", nf, 4 * nf, 3)
        conv(f"reconst_{i + 1}.final_conv.conv.1", nf, cfg["in_chans"], 3)
    return sd


# ----------------------------------------------------------------------------
# ENLCN (dlib/models/network_enlcn.py): EDSR body with Efficient Non-Local Contrastive Attention blocks
def _enlca(sd: SD, pre: str, x: Tensor, res_scale: float) -> Tensor:
    """ENLCA.forward in evaluation mode (network_enlcn.py:330-366; the contrastive term of training mode is dropped by
    ENLCN.forward :434-437) with ENLA / softmax_kernel / linear_attention (:207-300): 1x1 embeddings, L2-normalised and
    scaled by sqrt(6), positive random features exp(<x, w_j> - |x|^2 / 2) + 1e-4 against the stored projection matrix
    (a BUFFER of the state_dict: gaussian_orthogonal_random_matrix :52-81), then linear attention."""
    q = F.conv2d(x, sd[pre + ".conv_match1.0.weight"], sd[pre + ".conv_match1.0.bias"])
    k = F.conv2d(x, sd[pre + ".conv_match2.0.weight"], sd[pre + ".conv_match2.0.bias"])
    v = F.conv2d(x, sd[pre + ".conv_assembly.0.weight"], sd[pre + ".conv_assembly.0.bias"])
    kk = math.sqrt(6)
    k = F.normalize(k, p=2, dim=1, eps=5e-5) * kk
    q = F.normalize(q, p=2, dim=1, eps=5e-5) * kk
    N, C, H, W = q.shape
    q = q.permute(0, 2, 3, 1).reshape(N, 1, H * W, C)
    k = k.permute(0, 2, 3, 1).reshape(N, 1, H * W, C)
    v = v.permute(0, 2, 3, 1).reshape(N, 1, H * W, -1)
    P = sd[pre + ".attn_fn.projection_matrix"]
    ratio = P.shape[0] ** -0.5
    proj = P[None, None].expand(N, 1, -1, -1)

    def feat(d):
        dash = torch.einsum("...id,...jd->...ij", d, proj)
        diag = (torch.sum(d ** 2, dim=-1) / 2.0).unsqueeze(dim=-1)
        return ratio * (torch.exp(dash - diag) + 1e-4)
    q, k = feat(q), feat(k)
    d_inv = 1.0 / torch.einsum("...nd,...d->...n", q, k.sum(dim=-2))
    context = torch.einsum("...nd,...ne->...de", k, v)
    out = torch.einsum("...de,...nd,...n->...ne", context, q, d_inv).squeeze(1)
    return out.permute(0, 2, 1).reshape(N, -1, H, W) * res_scale + x


def enlcn_forward(sd: SD, x: Tensor, upscale: int, n_resblock: int = 32, res_scale: float = 0.1) -> Tensor:
    """ENLCN.forward (network_enlcn.py:421-448): head conv; body = ENLCA, then n_resblock ResBlocks (conv-ReLU-conv,
    x res_scale, + x: :139-160) with an ENLCA behind every eighth, then a conv; long skip; Upsampler (:163-196) + conv.
    sub_mean / add_mean are not applied (:423,441)."""
    x = F.conv2d(x, sd["head.0.weight"], sd["head.0.bias"], padding=1)
    res, i = x, 0
    res = _enlca(sd, f"body.{i}", res, res_scale); i += 1
    for b in range(n_resblock):
        r = F.conv2d(res, sd[f"body.{i}.body.0.weight"], sd[f"body.{i}.body.0.bias"], padding=1)
        r = F.conv2d(F.relu(r), sd[f"body.{i}.body.2.weight"], sd[f"body.{i}.body.2.bias"], padding=1)
        res = r * res_scale + res
        i += 1
        if (b + 1) % 8 == 0:
            res = _enlca(sd, f"body.{i}", res, res_scale); i += 1
    res = F.conv2d(res, sd[f"body.{i}.weight"], sd[f"body.{i}.bias"], padding=1) + x
    for st in range(int(math.log2(upscale))):
        res = F.pixel_shuffle(F.conv2d(res, sd[f"tail.0.{2 * st}.weight"], sd[f"tail.0.{2 * st}.bias"], padding=1), 2)
    return F.conv2d(res, sd["tail.1.weight"], sd["tail.1.bias"], padding=1)


def enlcn_init_state_dict(upscale: int, in_chans: int = 1, n_resblock: int = 32, n_feats: int = 256, seed: int = 0,
                          bias_std: float = 0.02, w_gain: float = 1.0) -> SD:
    """Seeded weights in the reference's state_dict layout and order (N(0, gain / sqrt(fan_in)); the frozen MeanShift convs
    as the reference builds them :26-37; the projection matrices as gaussian_orthogonal_random_matrix builds them, from
    this generator)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(pre, ci, co, k):
        sd[pre + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (w_gain / math.sqrt(ci * k * k))
        sd[pre + ".bias"] = torch.randn(co, generator=g) * bias_std

    def shift(pre, sign):
        sd[pre + ".weight"] = torch.eye(3).view(3, 3, 1, 1)
        sd[pre + ".bias"] = sign * torch.tensor([0.4488, 0.4371, 0.4040])

    def enlca(pre):
        d = n_feats // 4
        conv(pre + ".conv_match1.0", n_feats, d, 1)
        conv(pre + ".conv_match2.0", n_feats, d, 1)
        conv(pre + ".conv_assembly.0", n_feats, n_feats, 1)
        blocks = []
        for _ in range(128 // d):
            qm, _ = torch.linalg.qr(torch.randn(d, d, generator=g))
            blocks.append(qm.t())
        rem = 128 - (128 // d) * d
        if rem:
            qm, _ = torch.linalg.qr(torch.randn(d, d, generator=g))
            blocks.append(qm.t()[:rem])
        mult = torch.randn(128, d, generator=g).norm(dim=1)
        sd[pre + ".attn_fn.projection_matrix"] = torch.diag(mult) @ torch.cat(blocks)

    shift("sub_mean", -1.0)
    shift("add_mean", 1.0)
    conv("head.0", in_chans, n_feats, 3)
    i = 0
    enlca(f"body.{i}"); i += 1
    for b in range(n_resblock):
        conv(f"body.{i}.body.0", n_feats, n_feats, 3)
        conv(f"body.{i}.body.2", n_feats, n_feats, 3)
        i += 1
        if (b + 1) % 8 == 0:
            enlca(f"body.{i}"); i += 1
    conv(f"body.{i}", n_feats, n_feats, 3)
    for st in range(int(math.log2(upscale))):
        conv(f"tail.0.{2 * st}", n_feats, 4 * n_feats, 3)
    conv("tail.1", n_feats, in_chans, 3)
    return sd


# ----------------------------------------------------------------------------
# NLSN (dlib/models/network_nlsn.py): EDSR body with Non-Local Sparse Attention blocks
def _nlsa(sd: SD, pre: str, x: Tensor, n_hashes: int, chunk_size: int, res_scale: float, rotations=None,
          indices=None, taps=None) -> Tensor:
    """NonLocalSparseAttention.forward (network_nlsn.py:131-268).  The reference draws the LSH rotations with torch.randn at
    every call (:152-155) and orders the hash codes with an UNSTABLE sort (:201): both are arguments here -- rotations
    [1, C, n_hashes, hash_buckets // 2] (None: drawn from the global generator, exactly as the reference does), indices
    [N, n_hashes * L] (None: torch's sort, as the reference) -- so that another implementation's draw / tie order can be
    replayed.  taps (a dict) receives the hash codes and the indices used."""
    N, _, H, W = x.shape
    xe = F.conv2d(x, sd[pre + ".conv_match.0.weight"], sd[pre + ".conv_match.0.bias"], padding=1)
    ye = F.conv2d(x, sd[pre + ".conv_assembly.0.weight"], sd[pre + ".conv_assembly.0.bias"])
    x_embed = xe.view(N, -1, H * W).contiguous().permute(0, 2, 1)
    y_embed = ye.view(N, -1, H * W).contiguous().permute(0, 2, 1)
    L, C = x_embed.shape[-2:]
    Cy = y_embed.shape[-1]
    hash_buckets = min(L // chunk_size + (L // chunk_size) % 2, 128)
    if rotations is None:
        rotations = torch.randn((1, C, n_hashes, hash_buckets // 2), dtype=x.dtype)
    rot = rotations.expand(N, -1, -1, -1)
    rotated = torch.einsum('btf,bfhi->bhti', x_embed, rot)
    rotated = torch.cat([rotated, -rotated], dim=-1)
    codes = torch.argmax(rotated, dim=-1)
    offsets = torch.reshape(torch.arange(n_hashes) * hash_buckets, (1, -1, 1))
    codes = torch.reshape(codes + offsets, (N, -1,)).detach()
    if indices is None:
        _, indices = codes.sort(dim=-1)
    _, undo_sort = indices.sort(dim=-1)
    if taps is not None:
        taps["codes"], taps["indices"], taps["hash_buckets"] = codes, indices, hash_buckets
    mod_indices = indices % L

    def bsel(values, idx):
        return values.gather(1, idx[:, :, None].expand(-1, -1, values.shape[-1]))
    xs, ys = bsel(x_embed, mod_indices), bsel(y_embed, mod_indices)
    padding = chunk_size - L % chunk_size if L % chunk_size != 0 else 0
    xb = torch.reshape(xs, (N, n_hashes, -1, C))
    yb = torch.reshape(ys, (N, n_hashes, -1, Cy))
    if padding:
        xb = torch.cat([xb, xb[:, :, -padding:, :].clone()], dim=2)
        yb = torch.cat([yb, yb[:, :, -padding:, :].clone()], dim=2)
    xb = torch.reshape(xb, (N, n_hashes, -1, chunk_size, C))
    yb = torch.reshape(yb, (N, n_hashes, -1, chunk_size, Cy))
    xm = F.normalize(xb, p=2, dim=-1, eps=5e-5)

    def adj(t):
        back = torch.cat([t[:, :, -1:, ...], t[:, :, :-1, ...]], dim=2)
        fwd = torch.cat([t[:, :, 1:, ...], t[:, :, :1, ...]], dim=2)
        return torch.cat([t, back, fwd], dim=3)
    xm, yb = adj(xm), adj(yb)
    raw = torch.einsum('bhkie,bhkje->bhkij', xb, xm)
    bucket_score = torch.logsumexp(raw, dim=-1, keepdim=True)
    score = torch.exp(raw - bucket_score)
    bucket_score = torch.reshape(bucket_score, [N, n_hashes, -1])
    ret = torch.einsum('bukij,bukje->bukie', score, yb)
    ret = torch.reshape(ret, (N, n_hashes, -1, Cy))
    if padding:
        ret = ret[:, :, :-padding, :].clone()
        bucket_score = bucket_score[:, :, :-padding].clone()
    ret = torch.reshape(ret, (N, -1, Cy))
    bucket_score = torch.reshape(bucket_score, (N, -1,))
    ret = bsel(ret, undo_sort)
    bucket_score = bucket_score.gather(1, undo_sort)
    ret = torch.reshape(ret, (N, n_hashes, L, Cy))
    bucket_score = torch.reshape(bucket_score, (N, n_hashes, L, 1))
    probs = F.softmax(bucket_score, dim=1)
    ret = torch.sum(ret * probs, dim=1)
    return ret.permute(0, 2, 1).view(N, -1, H, W).contiguous() * res_scale + x


def nlsn_body_layout(n_resblocks: int):
    """(index in NLSN.body, kind) -- network_nlsn.py:326-341: an attention block first, one behind every eighth ResBlock."""
    out, i = [(0, "nlsa")], 1
    for b in range(n_resblocks):
        out.append((i, "res")); i += 1
        if (b + 1) % 8 == 0:
            out.append((i, "nlsa")); i += 1
    out.append((i, "conv"))
    return out


def nlsn_forward(sd: SD, x: Tensor, upscale: int, n_resblocks: int = 32, n_hashes: int = 4, chunk_size: int = 144,
                 res_scale: float = 0.1, rotations=None, indices=None, taps=None) -> Tensor:
    """NLSN.forward (network_nlsn.py:355-369); rotations / indices: one entry per attention block, in body order (None:
    the reference's own draws / sort); taps: list that receives one dict per attention block."""
    x = F.conv2d(x, sd["head.0.weight"], sd["head.0.bias"], padding=1)
    res, a = x, 0
    for i, kind in nlsn_body_layout(n_resblocks):
        if kind == "nlsa":
            tp = {} if taps is not None else None
            res = _nlsa(sd, f"body.{i}", res, n_hashes, chunk_size, res_scale,
                        None if rotations is None else rotations[a], None if indices is None else indices[a], tp)
            if taps is not None:
                taps.append(tp)
            a += 1
        elif kind == "res":
            r = F.conv2d(res, sd[f"body.{i}.body.0.weight"], sd[f"body.{i}.body.0.bias"], padding=1)
            r = F.conv2d(F.relu(r), sd[f"body.{i}.body.2.weight"], sd[f"body.{i}.body.2.bias"], padding=1)
            res = r * res_scale + res
        else:
            res = F.conv2d(res, sd[f"body.{i}.weight"], sd[f"body.{i}.bias"], padding=1)
    res = res + x
    for st in range(int(math.log2(upscale))):
        res = F.pixel_shuffle(F.conv2d(res, sd[f"tail.0.{2 * st}.weight"], sd[f"tail.0.{2 * st}.bias"], padding=1), 2)
    return F.conv2d(res, sd["tail.1.weight"], sd["tail.1.bias"], padding=1)


def nlsn_init_state_dict(upscale: int, in_chans: int = 1, n_resblocks: int = 32, n_feats: int = 256, seed: int = 0,
                         bias_std: float = 0.02, w_gain: float = 1.0) -> SD:
    """Seeded weights in the reference's state_dict layout and order (N(0, gain / sqrt(fan_in)); the frozen MeanShift
    convs as the reference builds them :42-53)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(pre, ci, co, k):
        sd[pre + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (w_gain / math.sqrt(ci * k * k))
        sd[pre + ".bias"] = torch.randn(co, generator=g) * bias_std

    def shift(pre, sign):
        sd[pre + ".weight"] = torch.eye(3).view(3, 3, 1, 1)
        sd[pre + ".bias"] = sign * torch.tensor([0.4488, 0.4371, 0.4040])

    shift("sub_mean", -1.0)
    shift("add_mean", 1.0)
    conv("head.0", in_chans, n_feats, 3)
    for i, kind in nlsn_body_layout(n_resblocks):
        if kind == "nlsa":
            conv(f"body.{i}.conv_match.0", n_feats, n_feats // 4, 3)
            conv(f"body.{i}.conv_assembly.0", n_feats, n_feats, 1)
        elif kind == "res":
            conv(f"body.{i}.body.0", n_feats, n_feats, 3)
            conv(f"body.{i}.body.2", n_feats, n_feats, 3)
        else:
            conv(f"body.{i}", n_feats, n_feats, 3)
    for st in range(int(math.log2(upscale))):
        conv(f"tail.0.{2 * st}", n_feats, 4 * n_feats, 3)
    conv("tail.1", n_feats, in_chans, 3)
    return sd


# ----------------------------------------------------------------------------
# DFCAN (dlib/models/network_dfcan.py): Fourier channel attention
def _dfcan_fftshift2d(img: Tensor) -> Tensor:
    """fftshift2d (network_dfcan.py:27-36): the four quadrants swapped, split at h // 2, w // 2."""
    _, _, h, w = img.shape
    fs11, fs12 = img[:, :, h // 2:, w // 2:], img[:, :, h // 2:, :w // 2]
    fs21, fs22 = img[:, :, :h // 2, w // 2:], img[:, :, :h // 2, :w // 2]
    return torch.cat([torch.cat([fs11, fs21], axis=2), torch.cat([fs12, fs22], axis=2)], axis=3)


def _dfcan_rcab(sd: SD, pre: str, x: Tensor, gamma: float = 0.8) -> Tensor:
    """RCAB.forward (network_dfcan.py:39-70): two conv + GELU, then the channel gate from the spectrum's magnitude:
    |FFT2|^0.8 (+1e-8 inside the power), fftshift, conv + ReLU, global average, 64 -> 4 -> 64 with ReLU / sigmoid."""
    x0 = x
    x = F.gelu(F.conv2d(x, sd[pre + ".conv_gelu1.0.weight"], sd[pre + ".conv_gelu1.0.bias"], padding=1))
    x = F.gelu(F.conv2d(x, sd[pre + ".conv_gelu2.0.weight"], sd[pre + ".conv_gelu2.0.bias"], padding=1))
    x1 = x
    x = torch.fft.fftn(x, dim=(2, 3))
    x = torch.pow(torch.abs(x) + 1e-8, gamma)
    x = _dfcan_fftshift2d(x)
    x = F.relu(F.conv2d(x, sd[pre + ".conv_relu1.0.weight"], sd[pre + ".conv_relu1.0.bias"], padding=1))
    x = F.adaptive_avg_pool2d(x, 1)
    x = F.relu(F.conv2d(x, sd[pre + ".conv_relu2.0.weight"], sd[pre + ".conv_relu2.0.bias"]))
    x = torch.sigmoid(F.conv2d(x, sd[pre + ".conv_sigmoid.0.weight"], sd[pre + ".conv_sigmoid.0.bias"]))
    return x0 + x1 * x


def dfcan_forward(sd: SD, x: Tensor, upscale: int) -> Tensor:
    """DFCAN.forward (network_dfcan.py:86-116): conv + GELU; 4 residual groups of 4 RCABs; conv 64 -> 64 s^2 + GELU;
    PixelShuffle(s); conv + sigmoid."""
    x = F.gelu(F.conv2d(x, sd["input.0.weight"], sd["input.0.bias"], padding=1))
    for g in range(4):
        x0 = x
        for r in range(4):
            x = _dfcan_rcab(sd, f"RGs.{g}.RCABs.{r}", x)
        x = x0 + x
    x = F.gelu(F.conv2d(x, sd["conv_gelu.0.weight"], sd["conv_gelu.0.bias"], padding=1))
    x = F.pixel_shuffle(x, upscale)
    return torch.sigmoid(F.conv2d(x, sd["conv_sigmoid.0.weight"], sd["conv_sigmoid.0.bias"], padding=1))


def dfcan_init_state_dict(upscale: int, in_chans: int = 1, seed: int = 0, bias_std: float = 0.02) -> SD:
    """Seeded weights in the reference's state_dict layout and order (N(0, 1 / sqrt(fan_in)))."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(pre, ci, co, k):
        sd[pre + ".weight"] = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
        sd[pre + ".bias"] = torch.randn(co, generator=g) * bias_std
    conv("input.0", in_chans, 64, 3)
    for gi in range(4):
        for r in range(4):
            pre = f"RGs.{gi}.RCABs.{r}"
            conv(pre + ".conv_gelu1.0", 64, 64, 3)
            conv(pre + ".conv_gelu2.0", 64, 64, 3)
            conv(pre + ".conv_relu1.0", 64, 64, 3)
            conv(pre + ".conv_relu2.0", 64, 4, 1)
            conv(pre + ".conv_sigmoid.0", 4, 64, 1)
    conv("conv_gelu.0", 64, 64 * upscale ** 2, 3)
    conv("conv_sigmoid.0", 64, in_chans, 3)
    return sd


# ----------------------------------------------------------------------------
# ACT (dlib/models/network_act.py): CNN (RCAN) branch + transformer branch with cross-scale token attention, fused
def _act_ln(sd: SD, pre: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _act_heads(t: Tensor, h: int) -> Tensor:              # 'b n (h d) -> b h n d'
    b, n, hd = t.shape
    return t.reshape(b, n, h, hd // h).permute(0, 2, 1, 3)


def _act_attend(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    dots = torch.einsum('bhid,bhjd->bhij', q, k) * scale
    out = torch.einsum('bhij,bhjd->bhid', dots.softmax(dim=-1), v)
    b, h, n, d = out.shape
    return out.permute(0, 2, 1, 3).reshape(b, n, h * d)   # 'b h n d -> b n (h d)'


def _act_self_attention(sd: SD, pre: str, x: Tensor, heads: int, dim_head: int) -> Tensor:
    """PreNorm(SelfAttention) (network_act.py:115-183)"""
    xn = _act_ln(sd, pre + ".norm", x)
    q, k, v = F.linear(xn, sd[pre + ".fn.to_qkv.weight"]).chunk(3, dim=-1)
    out = _act_attend(_act_heads(q, heads), _act_heads(k, heads), _act_heads(v, heads), dim_head ** -0.5)
    return F.linear(out, sd[pre + ".fn.to_out.0.weight"], sd[pre + ".fn.to_out.0.bias"])


def _act_cross_attention(sd: SD, pre: str, xq: Tensor, xkv: Tensor, heads: int, dim_head: int) -> Tensor:
    """PreNorm2(CrossAttention) (network_act.py:125-227)"""
    q = F.linear(_act_ln(sd, pre + ".norm", xq), sd[pre + ".fn.to_q.weight"])
    k, v = F.linear(_act_ln(sd, pre + ".norm2", xkv), sd[pre + ".fn.to_kv.weight"]).chunk(2, dim=-1)
    out = _act_attend(_act_heads(q, heads), _act_heads(k, heads), _act_heads(v, heads), dim_head ** -0.5)
    return F.linear(out, sd[pre + ".fn.to_out.0.weight"], sd[pre + ".fn.to_out.0.bias"])


def _act_ffn(sd: SD, pre: str, x: Tensor) -> Tensor:
    """PreNorm(FeedForward) (:136-148)"""
    h = F.gelu(F.linear(_act_ln(sd, pre + ".norm", x), sd[pre + ".fn.net.0.weight"], sd[pre + ".fn.net.0.bias"]))
    return F.linear(h, sd[pre + ".fn.net.3.weight"], sd[pre + ".fn.net.3.bias"])


def _act_ln_mlp(sd: SD, pre: str, x: Tensor) -> Tensor:
    """Sequential(LayerNorm, Linear, GELU, Linear) (:396-402,416-421,450-456)"""
    h = F.gelu(F.linear(_act_ln(sd, pre + ".0", x), sd[pre + ".1.weight"], sd[pre + ".1.bias"]))
    return F.linear(h, sd[pre + ".3.weight"], sd[pre + ".3.bias"])


def _act_rcab(sd: SD, pre: str, x: Tensor) -> Tensor:
    """RCAB (:250-277): conv, ReLU, conv, CALayer (:230-247), + x"""
    r = F.conv2d(x, sd[pre + ".body.0.weight"], sd[pre + ".body.0.bias"], padding=1)
    r = F.conv2d(F.relu(r), sd[pre + ".body.2.weight"], sd[pre + ".body.2.bias"], padding=1)
    y = F.adaptive_avg_pool2d(r, 1)
    y = F.relu(F.conv2d(y, sd[pre + ".body.3.conv_du.0.weight"], sd[pre + ".body.3.conv_du.0.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[pre + ".body.3.conv_du.2.weight"], sd[pre + ".body.3.conv_du.2.bias"]))
    return r * y + x


def act_forward(sd: SD, x: Tensor, upscale: int, n_feats: int = 64, n_resblocks: int = 12, n_heads: int = 8,
                n_fusionblocks: int = 4, token_size: int = 3, taps=None) -> Tensor:
    """ACT.forward (network_act.py:468-541), evaluation mode (dropout 0)."""
    h, w = x.shape[-2:]
    ts = token_size
    emb = n_feats * ts * ts
    dim_head = emb // n_heads
    x = F.conv2d(x, sd["head.0.weight"], sd["head.0.bias"], padding=1)
    for j in (1, 2):                                       # two 5 x 5 ResBlocks (:50-76)
        r = F.conv2d(x, sd[f"head.{j}.body.0.weight"], sd[f"head.{j}.body.0.bias"], padding=2)
        r = F.conv2d(F.relu(r), sd[f"head.{j}.body.2.weight"], sd[f"head.{j}.body.2.bias"], padding=2)
        x = r + x
    identity = x

    def tap(name, v):
        if taps is not None:
            taps[name] = v
    tap("head", x)
    tk = F.unfold(x, ts, stride=ts).permute(0, 2, 1)       # 'b d t -> b t d'
    tk = F.linear(tk, sd["linear_encoding.weight"], sd["linear_encoding.bias"]) + tk
    tap("enc", tk)
    f = None
    for i in range(n_fusionblocks):
        tk = _act_self_attention(sd, f"mhsa_block.{i}.0", tk, n_heads, dim_head) + tk
        tap(f"sa{i}", tk)
        tk = _act_ffn(sd, f"mhsa_block.{i}.1", tk) + tk
        tap(f"ffn{i}", tk)
        ta, tb = torch.split(tk, emb // 2, -1)
        tb = F.fold(tb.permute(0, 2, 1), (h, w), ts, stride=ts)
        tb = F.unfold(tb, ts * 2, stride=ts).permute(0, 2, 1)
        tb = _act_ln_mlp(sd, f"csta_block.{i}.0", tb)
        _ta, _tb = ta, tb
        ta = _act_cross_attention(sd, f"csta_block.{i}.1", ta, _tb, n_heads // 2, dim_head) + ta
        tb = _act_cross_attention(sd, f"csta_block.{i}.2", tb, _ta, n_heads // 2, dim_head) + tb
        tb = _act_ln_mlp(sd, f"csta_block.{i}.3", tb)
        tb = F.fold(tb.permute(0, 2, 1), (h, w), ts * 2, stride=ts)
        tb = F.unfold(tb, ts, stride=ts).permute(0, 2, 1)
        tk = torch.cat((ta, tb), -1)
        tk = _act_ffn(sd, f"csta_block.{i}.4", tk) + tk
        tap(f"csta{i}", tk)
        x0 = x                                             # ResidualGroup (:280-301)
        for r in range(n_resblocks):
            x = _act_rcab(sd, f"cnn_branch.{i}.body.{r}", x)
        x = F.conv2d(x, sd[f"cnn_branch.{i}.body.{n_resblocks}.weight"], sd[f"cnn_branch.{i}.body.{n_resblocks}.bias"],
                     padding=1) + x0
        tap(f"cnn{i}", x)
        tk_res, x_res = tk, x
        tkimg = F.fold(tk.permute(0, 2, 1), (h, w), ts, stride=ts)
        f = torch.cat((x, tkimg), 1)
        g = f
        for j in range(4):                                 # FB (:304-318): 1 x 1 conv, ReLU, 1 x 1 conv (no bias), + input
            r = F.conv2d(F.relu(F.conv2d(g, sd[f"fusion_block.{i}.{j}.body.0.weight"])), sd[f"fusion_block.{i}.{j}.body.2.weight"])
            g = r + g
        f = f + g
        tap(f"f{i}", f)
        if i != n_fusionblocks - 1:
            tkimg, x = torch.split(f, n_feats, 1)
            tk = F.unfold(tkimg, ts, stride=ts).permute(0, 2, 1)
            tk = _act_ln_mlp(sd, f"fusion_mlp.{i}", tk) + tk_res
            x = F.conv2d(F.relu(F.conv2d(x, sd[f"fusion_cnn.{i}.0.weight"], sd[f"fusion_cnn.{i}.0.bias"], padding=1)),
                         sd[f"fusion_cnn.{i}.2.weight"], sd[f"fusion_cnn.{i}.2.bias"], padding=1) + x_res
    x = F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1) + identity
    for st in range(int(math.log2(upscale))):
        x = F.pixel_shuffle(F.conv2d(x, sd[f"tail.0.{2 * st}.weight"], sd[f"tail.0.{2 * st}.bias"], padding=1), 2)
    return F.conv2d(x, sd["tail.1.weight"], sd["tail.1.bias"], padding=1)


# ----------------------------------------------------------------------------
# OmniSR (dlib/models/network_omni_sr.py): omni self-attention blocks (window / grid attention, spatial and channel)
def _omni_rel_pos_indices(w: int) -> Tensor:
    """Attention.__init__ (network_omni_sr.py:243-255)"""
    pos = torch.arange(w)
    grid = torch.stack(torch.meshgrid(pos, pos, indexing="ij")).reshape(2, -1).t()      # '(i j) c'
    rel = grid[:, None, :] - grid[None, :, :] + (w - 1)
    return (rel * torch.tensor([2 * w - 1, 1])).sum(dim=-1)


def _omni_ln2d(sd: SD, pre: str, x: Tensor) -> Tensor:
    """LayerNorm2d (:27-65): over the channels of every pixel, eps 1e-6"""
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + 1e-6).sqrt()
    return sd[pre + ".weight"].view(1, -1, 1, 1) * y + sd[pre + ".bias"].view(1, -1, 1, 1)


def _omni_mbconv(sd: SD, pre: str, x: Tensor) -> Tensor:
    """MBConv(expansion 1, no downsample) wrapped in MBConvResidual (:151-189): 1x1, GELU, depthwise 3x3, GELU,
    squeeze-excitation (mean, Linear, SiLU, Linear, sigmoid; :133-148), 1x1, + x"""
    h = F.gelu(F.conv2d(x, sd[pre + ".fn.0.weight"], sd[pre + ".fn.0.bias"]))
    h = F.gelu(F.conv2d(h, sd[pre + ".fn.2.weight"], sd[pre + ".fn.2.bias"], padding=1, groups=h.shape[1]))
    g = h.mean(dim=(2, 3))
    g = torch.sigmoid(F.linear(F.silu(F.linear(g, sd[pre + ".fn.4.gate.1.weight"])), sd[pre + ".fn.4.gate.3.weight"]))
    h = h * g[:, :, None, None]
    return F.conv2d(h, sd[pre + ".fn.5.weight"], sd[pre + ".fn.5.bias"]) + x


def _omni_attention(sd: SD, pre: str, x: Tensor, w: int, pe: bool) -> Tensor:
    """PreNormResidual(Attention) on 'b x y w1 w2 d' (:192-306): LayerNorm, qkv without bias, heads of dim / 4, scaled
    dot product + relative-position bias, softmax, output Linear without bias, + x"""
    b, X, Y, w1, w2, d = x.shape
    heads = 4                                              # dim_head = channel_num // 4 (:447)
    t = F.layer_norm(x, (d,), sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], 1e-5).reshape(b * X * Y, w1 * w2, d)
    q, k, v = F.linear(t, sd[pre + ".fn.to_qkv.weight"]).chunk(3, dim=-1)

    def hd(u):                                             # 'b n (h d) -> b h n d'
        return u.reshape(u.shape[0], u.shape[1], heads, d // heads).permute(0, 2, 1, 3)
    q, k, v = hd(q), hd(k), hd(v)
    q = q * (d // heads) ** -0.5
    sim = torch.einsum('bhid,bhjd->bhij', q, k)
    if pe:
        bias = sd[pre + ".fn.rel_pos_bias.weight"][_omni_rel_pos_indices(w)]       # [64, 64, heads]
        sim = sim + bias.permute(2, 0, 1)
    out = torch.einsum('bhij,bhjd->bhid', sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b * X * Y, w1, w2, d)                    # 'b h (w1 w2) d -> b w1 w2 (h d)'
    out = F.linear(out, sd[pre + ".fn.to_out.0.weight"])
    return out.reshape(b, X, Y, w1, w2, d) + x


def _omni_ffn(sd: SD, pre: str, x: Tensor) -> Tensor:
    """Conv_PreNormResidual(Gated_Conv_FeedForward) (:202-209,308-329): LayerNorm2d, 1x1 C -> 2C, depthwise 3x3,
    gelu(x1) * x2, 1x1, + x (no biases)"""
    h = F.conv2d(_omni_ln2d(sd, pre + ".norm", x), sd[pre + ".fn.project_in.weight"])
    h = F.conv2d(h, sd[pre + ".fn.dwconv.weight"], padding=1, groups=h.shape[1])
    x1, x2 = h.chunk(2, dim=1)
    return F.conv2d(F.gelu(x1) * x2, sd[pre + ".fn.project_out.weight"]) + x


def _omni_channel_attention(sd: SD, pre: str, x: Tensor, ps: int, grid: bool) -> Tensor:
    """Conv_PreNormResidual(Channel_Attention / Channel_Attention_grid) (:332-428): LayerNorm2d, qkv = depthwise 3x3 of a
    1x1, per (window, head) -- grid: per (in-window position, head) -- the d x d attention between L2-normalised channel
    vectors, times the head's temperature, softmax, times v; 1x1; + x"""
    from einops import rearrange
    b, c, h, w = x.shape
    heads = 4
    n = _omni_ln2d(sd, pre + ".norm", x)
    qkv = F.conv2d(n, sd[pre + ".fn.qkv.weight"])
    qkv = F.conv2d(qkv, sd[pre + ".fn.qkv_dwconv.weight"], padding=1, groups=3 * c).chunk(3, dim=1)
    pat = ('b (head d) (h ph) (w pw) -> b (ph pw) head d (h w)' if grid else
           'b (head d) (h ph) (w pw) -> b (h w) head d (ph pw)')
    q, k, v = (rearrange(t, pat, ph=ps, pw=ps, head=heads) for t in qkv)
    q, k = F.normalize(q, dim=-1), F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * sd[pre + ".fn.temperature"]).softmax(dim=-1)
    out = attn @ v
    back = ('b (ph pw) head d (h w) -> b (head d) (h ph) (w pw)' if grid else
            'b (h w) head d (ph pw) -> b (head d) (h ph) (w pw)')
    out = rearrange(out, back, h=h // ps, w=w // ps, ph=ps, pw=ps, head=heads)
    return F.conv2d(out, sd[pre + ".fn.project_out.weight"]) + x


def _omni_esa(sd: SD, pre: str, x: Tensor) -> Tensor:
    """ESA (:85-114)"""
    c1_ = F.conv2d(x, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"])
    c1 = F.conv2d(c1_, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], stride=2)
    v_max = F.max_pool2d(c1, kernel_size=7, stride=3)
    c3 = F.conv2d(v_max, sd[pre + ".conv3.weight"], sd[pre + ".conv3.bias"], padding=1)
    c3 = F.interpolate(c3, (x.size(2), x.size(3)), mode='bilinear', align_corners=False)
    cf = F.conv2d(c1_, sd[pre + ".conv_f.weight"], sd[pre + ".conv_f.bias"])
    c4 = F.conv2d(c3 + cf, sd[pre + ".conv4.weight"], sd[pre + ".conv4.bias"])
    return x * torch.sigmoid(c4)


def omnisr_forward(sd: SD, x: Tensor, upscale: int, res_num: int = 5, block_num: int = 4, window_size: int = 8,
                   pe: bool = True, taps=None) -> Tensor:
    """OmniSR.forward (network_omni_sr.py:577-591) with OSAG (:495-524) and OSA_Block (:430-492)."""
    from einops import rearrange
    H, W = x.shape[2:]
    ws = window_size
    x = F.pad(x, (0, (ws - W % ws) % ws, 0, (ws - H % ws) % ws), 'constant', 0)
    residual = F.conv2d(x, sd["input.weight"], sd["input.bias"], padding=1)

    def tap(name, v):
        if taps is not None:
            taps[name] = v
    tap("input", residual)
    out = residual
    for g in range(res_num):
        gin = out
        for bk in range(block_num):
            pre = f"residual_layer.{g}.residual_layer.{bk}.layer"
            out = _omni_mbconv(sd, pre + ".0", out)
            tap(f"g{g}b{bk}.mb", out)
            t = rearrange(out, 'b d (x w1) (y w2) -> b x y w1 w2 d', w1=ws, w2=ws)
            t = _omni_attention(sd, pre + ".2", t, ws, pe)
            out = rearrange(t, 'b x y w1 w2 d -> b d (x w1) (y w2)')
            tap(f"g{g}b{bk}.att1", out)
            out = _omni_ffn(sd, pre + ".4", out)
            tap(f"g{g}b{bk}.ffn1", out)
            out = _omni_channel_attention(sd, pre + ".5", out, ws, False)
            tap(f"g{g}b{bk}.ca1", out)
            out = _omni_ffn(sd, pre + ".6", out)
            t = rearrange(out, 'b d (w1 x) (w2 y) -> b x y w1 w2 d', w1=ws, w2=ws)
            t = _omni_attention(sd, pre + ".8", t, ws, pe)
            out = rearrange(t, 'b x y w1 w2 d -> b d (w1 x) (w2 y)')
            tap(f"g{g}b{bk}.att2", out)
            out = _omni_ffn(sd, pre + ".10", out)
            out = _omni_channel_attention(sd, pre + ".11", out, ws, True)
            tap(f"g{g}b{bk}.ca2", out)
            out = _omni_ffn(sd, pre + ".12", out)
            tap(f"g{g}b{bk}.out", out)
        pre = f"residual_layer.{g}"
        out = F.conv2d(out, sd[pre + f".residual_layer.{block_num}.weight"], sd[pre + f".residual_layer.{block_num}.bias"]) + gin
        out = _omni_esa(sd, pre + ".esa", out)
        tap(f"g{g}.esa", out)
    out = F.conv2d(out, sd["output.weight"], sd["output.bias"], padding=1) + residual
    out = F.pixel_shuffle(F.conv2d(out, sd["up.0.weight"], sd["up.0.bias"], padding=1), upscale)
    return out[:, :, :H * upscale, :W * upscale]


# ----------------------------------------------------------------------------
# GRL (dlib/models/network_grl.py)
# ----------------------------------------------------------------------------
def _grl_partition(x: Tensor, ws) -> Tensor:
    """window_partition (network_grl.py:512-529): (B, H, W, C) -> (B nW, wh, ww, C)"""
    B, H, W, C = x.shape
    x = x.view(B, H // ws[0], ws[0], W // ws[1], ws[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws[0], ws[1], C)


def _grl_reverse(w: Tensor, ws, size) -> Tensor:
    """window_reverse (network_grl.py:744-760)"""
    H, W = size
    B = int(w.shape[0] / (H * W / ws[0] / ws[1]))
    x = w.view(B, H // ws[0], W // ws[1], ws[0], ws[1], -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def _grl_coords(n) -> Tensor:
    """_get_meshgrid_coords from (0, 0) (network_grl.py:1534-1539)"""
    c = torch.stack(torch.meshgrid([torch.arange(0, n[0]), torch.arange(0, n[1])], indexing="ij"))
    return torch.flatten(c, 1)


def grl_relative_position_index(ws, df: int = 1, window_to_anchor: bool = True) -> Tensor:
    """get_relative_position_index_simple (network_grl.py:1553-1577) with coords_diff_odd (:1542-1550): pairwise
    coordinate differences between the tokens of a window and of its anchor window (sides ws // df), shifted to start at 0
    and flattened with row stride aws[1] + ws[1] - 1."""
    aws = [w // df for w in ws]
    c, ca = _grl_coords(ws), _grl_coords(aws)
    width = aws[1] + ws[1] - 1
    if window_to_anchor:
        a, b, off = c, ca, [w - 1 for w in aws]
    else:
        a, b, off = ca, c, [w - 1 for w in ws]
    d = (a[:, :, None] - b[:, None, :]).permute(1, 2, 0).contiguous()
    d[:, :, 0] += off[0]
    d[:, :, 1] += off[1]
    d[:, :, 0] *= width
    return d.sum(-1)


def grl_coords_table(ws, df: int = 1) -> Tensor:
    """get_relative_coords_table_all (network_grl.py:1651-1699), pretrained sizes 0: the signed-log relative coordinates
    the CPB MLP reads, (1, wh + awh - 1, ww + aww - 1, 2)."""
    aws = [w // df for w in ws]
    ts_p = [w1 - 1 - (w1 - w2) // 2 for w1, w2 in zip(ws, aws)]
    ts_n = [-(w2 - 1) - (w1 - w2) // 2 for w1, w2 in zip(ws, aws)]
    ch = torch.arange(ts_n[0], ts_p[0] + 1, dtype=torch.float32)
    cw = torch.arange(ts_n[1], ts_p[1] + 1, dtype=torch.float32)
    t = torch.stack(torch.meshgrid([ch, cw], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    t[:, :, :, 0] /= ts_p[0]
    t[:, :, :, 1] /= ts_p[1]
    t *= 8
    return torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)


def _grl_fill(res, ws, shift) -> Tensor:
    """_fill_window (network_grl.py:1580-1604): the 3 x 3 region ids of a shifted partition, per window"""
    m = torch.zeros((1, *res, 1))
    cnt = 0
    for h in (slice(0, -ws[0]), slice(-ws[0], -shift[0]), slice(-shift[0], None)):
        for w in (slice(0, -ws[1]), slice(-ws[1], -shift[1]), slice(-shift[1], None)):
            m[:, h, w, :] = cnt
            cnt += 1
    return _grl_partition(m, ws).view(-1, ws[0] * ws[1])


def _grl_mask_fill(d: Tensor) -> Tensor:
    return d.masked_fill(d != 0, float(-100.0)).masked_fill(d == 0, float(0.0))


def grl_mask(res, ws, shift) -> Tensor:
    """calculate_mask (network_grl.py:1607-1622)"""
    m = _grl_fill(res, ws, shift)
    return _grl_mask_fill(m.unsqueeze(1) - m.unsqueeze(2))


def grl_mask_all(res, ws, shift, df: int = 1, window_to_anchor: bool = True) -> Tensor:
    """calculate_mask_all (network_grl.py:1625-1648)"""
    mw = _grl_fill(res, ws, shift)
    ma = _grl_fill([s // df for s in res], [s // df for s in ws], [s // df for s in shift])
    d = mw.unsqueeze(2) - ma.unsqueeze(1) if window_to_anchor else ma.unsqueeze(2) - mw.unsqueeze(1)
    return _grl_mask_fill(d)


def grl_buffers(x_size, window_size: int = 8, stripe_size=(8, 8), df: int = 2) -> SD:
    """GRL.set_table_index_mask (network_grl.py:1332-1375), stripe_groups (None, None): the 13 buffers the module registers
    (and recomputes for another input size), in registration order."""
    ws = [window_size, window_size]
    ss = list(stripe_size)
    sss = [s // 2 for s in ss]                      # _get_stripe_info(.., True, ..) :322-331
    rs = list(x_size)
    return {
        "table_w": grl_coords_table(ws), "table_sh": grl_coords_table(ss, df), "table_sv": grl_coords_table(ss[::-1], df),
        "index_w": grl_relative_position_index(ws),
        "index_sh_a2w": grl_relative_position_index(ss, df, False), "index_sh_w2a": grl_relative_position_index(ss, df, True),
        "index_sv_a2w": grl_relative_position_index(ss[::-1], df, False),
        "index_sv_w2a": grl_relative_position_index(ss[::-1], df, True),
        "mask_w": grl_mask(rs, ws, [w // 2 for w in ws]),
        "mask_sh_a2w": grl_mask_all(rs, ss, sss, df, False), "mask_sh_w2a": grl_mask_all(rs, ss, sss, df, True),
        "mask_sv_a2w": grl_mask_all(rs, ss[::-1], sss[::-1], df, False),
        "mask_sv_w2a": grl_mask_all(rs, ss[::-1], sss[::-1], df, True),
    }


def _grl_attn(sd: SD, pre: str, q: Tensor, k: Tensor, v: Tensor, table: Tensor, index: Tensor, mask: Optional[Tensor],
              reshape: bool = True) -> Tensor:
    """Attention.attn (network_grl.py:338-355) with AffineTransform.forward (:296-319): cosine similarity, the clamped
    per-head logit scale, 16 sigmoid(CPB MLP(table))[index] as the bias, the shift mask, softmax, @ v."""
    B_, _, H, hd = q.shape
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    _, nh, N1, N2 = attn.shape
    attn = attn * torch.clamp(sd[pre + ".logit_scale"], max=math.log(1.0 / 0.01)).exp()
    table = table.to(q.dtype)                     # (the float64 runs of the tests)
    bt = F.linear(F.relu(F.linear(table, sd[pre + ".cpb_mlp.0.weight"], sd[pre + ".cpb_mlp.0.bias"])),
                  sd[pre + ".cpb_mlp.2.weight"]).view(-1, nh)
    bias = bt[index.view(-1)].view(N1, N2, -1).permute(2, 0, 1).contiguous()
    attn = attn + (16 * torch.sigmoid(bias)).unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nh, N1, N2) + mask.to(attn.dtype).unsqueeze(1).unsqueeze(0)).view(-1, nh, N1, N2)
    x = attn.softmax(-1) @ v
    if reshape:
        x = x.transpose(1, 2).reshape(B_, -1, H * hd)
    return x


def _grl_block(sd: SD, pre: str, x: Tensor, x_size, buf: SD, i: int, heads_w: int, heads_s: int, ws, ss, df: int,
               tap=None) -> Tensor:
    """EfficientMixAttnTransformerBlock.forward (network_grl.py:1061-1076) around MixedAttention.forward (:861-887),
    WindowAttention.forward (:381-412), AnchorStripeAttention.forward (:463-514) without the stripe shift, CAB (:729-741),
    Mlp (:779-785).  Block i of a stage: shifted windows on even i, 'H' stripes on even i and 'W' (transposed sizes) on
    odd i (TransformerStage :141-143)."""
    B, L, C = x.shape
    H, W = x_size
    shift = ws[0] // 2 if i % 2 == 0 else 0
    st = "h" if i % 2 == 0 else "v"
    ssz = list(ss) if i % 2 == 0 else list(ss)[::-1]
    a = pre + ".attn"
    qkv = F.linear(x, sd[a + ".qkv.body.weight"], sd[a + ".qkv.body.bias"])
    qkv_w, qkv_s = torch.split(qkv, C * 3 // 2, dim=-1)
    # AnchorLinear (:611-620)
    t = F.avg_pool2d(x.transpose(1, 2).view(B, C, H, W), df, df).flatten(2).transpose(1, 2)
    anchor = F.linear(t, sd[a + ".anchor.body.0.reduction.weight"], sd[a + ".anchor.body.0.reduction.bias"])
    anchor = anchor.view(B, H // df, W // df, C // 2)
    # window attention
    q = qkv_w.view(B, H, W, -1)
    if shift > 0:
        q = torch.roll(q, shifts=(-shift, -shift), dims=(1, 2))
    q = _grl_partition(q, ws).view(-1, ws[0] * ws[1], C * 3 // 2)
    B_, N, _ = q.shape
    q = q.reshape(B_, N, 3, heads_w, -1).permute(2, 0, 3, 1, 4)
    xw = _grl_attn(sd, a + ".window_attn.attn_transform", q[0], q[1], q[2], buf["table_w"], buf["index_w"],
                   buf["mask_w"] if shift > 0 else None)
    xw = _grl_reverse(xw.view(-1, *ws, C // 2), ws, x_size)
    if shift > 0:
        xw = torch.roll(xw, shifts=(shift, shift), dims=(1, 2))
    xw = xw.view(B, L, C // 2)
    # anchored stripe attention
    asz = [s // df for s in ssz]
    q = _grl_partition(qkv_s.view(B, H, W, -1), ssz).view(-1, ssz[0] * ssz[1], C * 3 // 2)
    an = _grl_partition(anchor, asz).view(-1, asz[0] * asz[1], C // 2)
    B_, N1, _ = q.shape
    N2 = an.shape[1]
    q = q.reshape(B_, N1, 3, heads_s, -1).permute(2, 0, 3, 1, 4)
    an = an.reshape(B_, N2, heads_s, -1).permute(0, 2, 1, 3)
    xs = _grl_attn(sd, a + ".stripe_attn.attn_transform1", an, q[1], q[2], buf["table_s" + st], buf[f"index_s{st}_a2w"],
                   None, False)
    xs = _grl_attn(sd, a + ".stripe_attn.attn_transform2", q[0], an, xs, buf["table_s" + st], buf[f"index_s{st}_w2a"], None)
    xs = _grl_reverse(xs.view(B_, *ssz, C // 2), ssz, x_size).view(B, H * W, C // 2)
    att = F.linear(torch.cat([xw, xs], dim=-1), sd[a + ".proj.weight"], sd[a + ".proj.bias"])
    if tap is not None:
        tap(pre + ".attn", att)
    # CAB
    c = F.conv2d(x.transpose(1, 2).view(B, C, H, W).contiguous(), sd[pre + ".conv.cab.0.weight"], sd[pre + ".conv.cab.0.bias"],
                 padding=1)
    c = F.conv2d(F.gelu(c), sd[pre + ".conv.cab.2.weight"], sd[pre + ".conv.cab.2.bias"], padding=1)
    y = F.adaptive_avg_pool2d(c, 1)
    y = F.relu(F.conv2d(y, sd[pre + ".conv.cab.3.attention.1.weight"], sd[pre + ".conv.cab.3.attention.1.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[pre + ".conv.cab.3.attention.3.weight"], sd[pre + ".conv.cab.3.attention.3.bias"]))
    cab = (c * y).flatten(2).transpose(1, 2)
    if tap is not None:
        tap(pre + ".cab", cab)
    x = x + 1.0 * F.layer_norm(att, (C,), sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"]) + cab
    m = F.linear(F.gelu(F.linear(x, sd[pre + ".mlp.fc1.weight"], sd[pre + ".mlp.fc1.bias"])),
                 sd[pre + ".mlp.fc2.weight"], sd[pre + ".mlp.fc2.bias"])
    return x + 1.0 * F.layer_norm(m, (C,), sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"])


def grl_forward(sd: SD, x: Tensor, upscale: int, depths: Sequence[int] = (4, 4, 8, 8, 8, 4, 4), heads_w: int = 3,
                heads_s: int = 3, window_size: int = 8, stripe_size=(8, 8), df: int = 2, taps=None) -> Tensor:
    """GRL.forward (network_grl.py:1462-1512), upsampler 'pixelshuffle', 1 input channel (mean 0, img_range 1), the
    registry's options (select_network.py:70-90): linear qkv / output projections, avgpool anchors, '1conv' stage ends,
    local connection on, no stripe shift.  Inputs are reflect-padded to a multiple of the window (:1415-1424), the output is
    cropped.  The tables, indices and masks are those of the padded size (get_table_index_mask :1377-1398)."""
    H0, W0 = x.shape[2:]
    pad = max(window_size, max(stripe_size))
    x = F.pad(x, (0, (pad - W0 % pad) % pad, 0, (pad - H0 % pad) % pad), "reflect")
    x = (x - 0.0) * 1.0
    ws = [window_size, window_size]

    def tap(name, v):
        if taps is not None:
            taps[name] = v
    f0 = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    x_size = (f0.shape[2], f0.shape[3])
    B, C = f0.shape[:2]
    buf = grl_buffers(x_size, window_size, stripe_size, df)
    t = F.layer_norm(f0.flatten(2).transpose(1, 2), (C,), sd["norm_start.weight"], sd["norm_start.bias"])
    tap("start", t)
    for s, depth in enumerate(depths):
        res = t
        for i in range(depth):
            res = _grl_block(sd, f"layers.{s}.blocks.{i}", res, x_size, buf, i, heads_w, heads_s, ws, stripe_size, df, tap)
            tap(f"layers.{s}.blocks.{i}", res)
        res = F.conv2d(res.transpose(1, 2).view(B, C, *x_size), sd[f"layers.{s}.conv.weight"], sd[f"layers.{s}.conv.bias"],
                       padding=1).flatten(2).transpose(1, 2)
        t = res + t
        tap(f"layers.{s}", t)
    t = F.layer_norm(t, (C,), sd["norm_end.weight"], sd["norm_end.bias"]).transpose(1, 2).view(B, C, *x_size)
    f = F.conv2d(t, sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1) + f0
    tap("body", f)
    u = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
    k = 0
    while f"upsample.up.{k}.weight" in sd:
        u = F.pixel_shuffle(F.conv2d(u, sd[f"upsample.up.{k}.weight"], sd[f"upsample.up.{k}.bias"], padding=1), 2)
        k += 2
    y = F.conv2d(u, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    y = y / 1.0 + 0.0
    return y[:, :, :H0 * upscale, :W0 * upscale]


def grl_state_dict(layout, seed: int, img_size) -> SD:
    """seeded_state_dict over GRL's layout: the 13 registered buffers keep the values the reference computes for img_size,
    the logit scales are drawn around the reference's initial log 10 and one is pushed over the clamp at log 100."""
    fixed = grl_buffers((img_size, img_size) if isinstance(img_size, int) else img_size)
    sd = seeded_state_dict([kv for kv in layout if kv[0] not in fixed], seed)
    first = True
    for k in sd:
        if k.endswith("logit_scale"):
            sd[k] = sd[k] * 0.5 + math.log(10.0)
            if first:
                sd[k][0] = 5.0
                first = False
    out: SD = {}
    for k, _ in layout:
        out[k] = fixed[k] if k in fixed else sd[k]
    return out


def seeded_state_dict(layout, seed: int, bias_std: float = 0.02) -> SD:
    """Seeded weights for a (key, shape) layout taken from a module's own state_dict (ACT: 660 entries): matrices / conv
    kernels N(0, 1 / sqrt(fan_in)), LayerNorm-like scale vectors 1 + N(0, 0.1), other vectors N(0, bias_std); the frozen
    MeanShift convs keep the values the reference builds."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for k, shape in layout:
        shape = tuple(int(v) for v in shape)
        if k.startswith(("sub_mean", "add_mean")):
            sd[k] = (torch.eye(3).view(3, 3, 1, 1) if k.endswith("weight")
                     else (-1.0 if k.startswith("sub") else 1.0) * torch.tensor([0.4488, 0.4371, 0.4040]))
        elif len(shape) > 1:
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            sd[k] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[k] = bias_std * torch.randn(shape, generator=g)
    return sd


# ----------------------------------------------------------------------------
# MSLapSRN (dlib/models/network_mslapsr.py)
# ----------------------------------------------------------------------------
def mslapsrn_forward(sd: SD, x: Tensor, upscale: int) -> Tuple[Tensor, List[Tensor]]:
    """network_mslapsr.py:134-174: conv1 + LeakyReLU(0.2); per octave: 10 x (conv + LeakyReLU), ConvTranspose2d(64, 64,
    4, 2, 1) + LeakyReLU -> features, conv 64 -> 1 on them + ConvTranspose2d(1, 1, 4, 2, 1) of the current image ->
    the octave's image.  Returns (last image, [images of the earlier octaves]) = (output, net.intermediate_outs)."""
    lr = lambda t: F.leaky_relu(t, 0.2)
    feat = lr(F.conv2d(x, sd["conv1.0.weight"], sd["conv1.0.bias"], padding=1))
    img, outs = x, []
    for o in range(int(math.log2(upscale))):
        a, b, c = (f"laplacian_pyramid_conv{3 * o + k}" for k in (1, 2, 3))
        for k in range(10):
            feat = lr(F.conv2d(feat, sd[f"{a}.{k}.cl.0.weight"], sd[f"{a}.{k}.cl.0.bias"], padding=1))
        feat = lr(F.conv_transpose2d(feat, sd[f"{a}.10.weight"], sd[f"{a}.10.bias"], stride=2, padding=1))
        img = F.conv_transpose2d(img, sd[f"{b}.weight"], sd[f"{b}.bias"], stride=2, padding=1) + \
            F.conv2d(feat, sd[f"{c}.weight"], sd[f"{c}.bias"], padding=1)
        outs.append(img)
    return outs[-1], outs[:-1]


def mslapsrn_loss(out: Tensor, inter: Sequence[Tensor], target: Tensor, kind: str = "l1") -> Tensor:
    """model_plain.py:277-314 (loss_mslaprs): the loss of the output plus the same loss of every intermediate image
    against the bicubically resized target (align_corners=True, clamped to [0, 1]), divided by the number of images."""
    f = loss_l1 if kind == "l1" else loss_l2
    total = f(out, target)
    for t in inter:
        gt = torch.clamp(F.interpolate(target, size=t.shape[2:], mode="bicubic", align_corners=True), 0.0, 1.0)
        total = total + f(t, gt)
    return total / (len(inter) + 1.0)


def mslapsrn_init_state_dict(upscale: int, seed: int = 0) -> SD:
    """Seeded weights in the shapes / key order of the reference net: Kaiming-normal (fan_out) convs and the bilinear
    4x4 filter on the transposed convs as in network_mslapsr.py:176-188 (a law the reference defines but does not
    call), here with 10 % noise on the transposed-conv weights and small random biases so that every tap and every
    bias matters in the parity fixtures."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(name, co, ci):
        sd[name + ".weight"] = torch.randn(co, ci, 3, 3, generator=g) * math.sqrt(2.0 / (9 * co))
        sd[name + ".bias"] = torch.randn(co, generator=g) * 0.05

    def convT(name, c):
        k = torch.tensor([0.25, 0.75, 0.75, 0.25])
        w = (k[:, None] * k[None, :]).expand(c, c, 4, 4) / max(1.0, c / 4.0)
        sd[name + ".weight"] = w * (1 + 0.1 * torch.randn(c, c, 4, 4, generator=g))
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.05

    conv("conv1.0", 64, 1)
    for o in range(int(math.log2(upscale))):
        a, b, c = (f"laplacian_pyramid_conv{3 * o + k}" for k in (1, 2, 3))
        for k in range(10):
            conv(f"{a}.{k}.cl.0", 64, 64)
        convT(f"{a}.10", 64)
        convT(b, 1)
        conv(c, 1, 64)
    return sd


def conv_transpose4x4s2_as_conv3x3(wT: Tensor) -> Tensor:
    """ConvTranspose2d(Ci, Co, 4, stride 2, padding 1) == PixelShuffle(2) o Conv2d(Ci, 4*Co, 3, padding 1): output pixel
    (2y + i, 2x + j) takes input pixels (y + dy, x + dx) with kernel tap (i + 1 - 2 dy, j + 1 - 2 dx) where that is
    inside the 4x4 kernel -- 2 x 2 of the 3 x 3 taps per sub-pixel.  wT [Ci, Co, 4, 4] -> [4*Co, Ci, 3, 3] with
    PixelShuffle's channel order co*4 + 2i + j."""
    ci, co = wT.shape[:2]
    w = wT.new_zeros(co, 4, ci, 3, 3)
    for i in range(2):
        for j in range(2):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    ky, kx = i + 1 - 2 * dy, j + 1 - 2 * dx
                    if 0 <= ky < 4 and 0 <= kx < 4:
                        w[:, 2 * i + j, :, dy + 1, dx + 1] = wT[:, :, ky, kx].t()
    return w.reshape(4 * co, ci, 3, 3)


def drrn_forward(sd: SD, x: Tensor, upscale: int, num_residual_units: int) -> Tensor:
    """network_drrn.py:22-126.  The reference's ReLUs are in place: the first ReLU of the first residual unit
    rectifies the block input itself, so the identity every unit adds is relu(conv1 output)."""
    xi = torch.clamp(F.interpolate(x, size=(upscale * x.shape[2], upscale * x.shape[3]), mode="bicubic",
                                   align_corners=False), 0.0, 1.0)
    x0 = F.relu(F.conv2d(F.relu(xi), sd["conv1.1.weight"], padding=1))
    out = x0
    for _ in range(num_residual_units):
        out = F.conv2d(F.relu(out), sd["trunk.residual_unit.1.weight"], padding=1)
        out = F.conv2d(F.relu(out), sd["trunk.residual_unit.3.weight"], padding=1)
        out = out + x0
    return F.conv2d(F.relu(out), sd["conv2.1.weight"], padding=1) + xi


def drrn_init_state_dict(in_chans: int = 1, seed: int = 0) -> SD:
    """network_drrn.py:128-136: kaiming_normal_(fan_out, relu) = N(0, sqrt(2 / (9 * Cout)))."""
    g = torch.Generator().manual_seed(seed)
    return {"conv1.1.weight": torch.randn(128, in_chans, 3, 3, generator=g) * math.sqrt(2 / (9 * 128)),
            "trunk.residual_unit.1.weight": torch.randn(128, 128, 3, 3, generator=g) * math.sqrt(2 / (9 * 128)),
            "trunk.residual_unit.3.weight": torch.randn(128, 128, 3, 3, generator=g) * math.sqrt(2 / (9 * 128)),
            "conv2.1.weight": torch.randn(in_chans, 128, 3, 3, generator=g) * math.sqrt(2 / (9 * in_chans))}


# ----------------------------------------------------------------------------
# MemNet (dlib/models/network_memnet.py)
# ----------------------------------------------------------------------------
def batchnorm2d(sd: SD, pre: str, x: Tensor, training: bool, stats: Optional[dict] = None, eps: float = 1e-5,
                momentum: float = 0.1) -> Tensor:
    """nn.BatchNorm2d as the reference's nets use it (network_memnet.py:28,31,60,101,113): training mode normalises
    with the batch mean and BIASED variance and moves the running statistics by ``momentum`` towards the batch mean /
    UNBIASED variance (written into ``stats``, keyed like the module's buffers); eval mode uses the running ones."""
    g, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if training:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        if stats is not None:
            n = x.numel() / x.shape[1]
            rm = stats.get(pre + ".running_mean", sd[pre + ".running_mean"])
            rv = stats.get(pre + ".running_var", sd[pre + ".running_var"])
            stats[pre + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach()
            stats[pre + ".running_var"] = (1 - momentum) * rv + momentum * var.detach() * (n / max(n - 1.0, 1.0))
            stats[pre + ".num_batches_tracked"] = stats.get(pre + ".num_batches_tracked",
                                                            sd[pre + ".num_batches_tracked"]) + 1
    else:
        mean, var = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    sh = (1, -1, 1, 1)
    return (x - mean.view(sh)) / torch.sqrt(var.view(sh) + eps) * g.view(sh) + b.view(sh)


def memnet_forward(sd: SD, x: Tensor, upscale: int, num_memory_blocks: int, num_residual_blocks: int,
                   training: bool = False, stats: Optional[dict] = None) -> Tensor:
    """network_memnet.py:143-167: bicubic up (align_corners False, clamped); BN-ReLU-conv 1 -> 64; memory blocks; BN-ReLU-
    conv1x1 64 -> 1; + the interpolated input.  A memory block (network_memnet.py:66-78) runs its WHOLE chain of
    residual units ``num_residual_blocks`` times (``self.recursive_unit(out)`` is the Sequential of all of them) and keeps
    the result of every pass as a short-term memory; the gate unit (BN-ReLU-conv1x1) reads the concatenation of those
    and of all long-term memories (the extractor's features and every earlier block's output).  A residual unit is
    x + conv(relu(BN(conv(relu(BN(x)))))) (:36-40).  No conv has a bias."""
    R = num_residual_blocks
    bn = lambda pre, t: F.relu(batchnorm2d(sd, pre, t, training, stats))
    xi = torch.clamp(F.interpolate(x, size=(upscale * x.shape[2], upscale * x.shape[3]), mode="bicubic",
                                   align_corners=False), 0.0, 1.0)
    out = F.conv2d(bn("feature_extractor.0", xi), sd["feature_extractor.2.weight"], padding=1)
    longs = [out]
    for i in range(num_memory_blocks):
        mb = f"dense_memory_blocks.{i}"
        shorts = []
        for _ in range(R):
            for j in range(R):
                ru = f"{mb}.recursive_unit.{j}.residual_block"
                t = F.conv2d(bn(ru + ".0", out), sd[ru + ".2.weight"], padding=1)
                out = F.conv2d(bn(ru + ".3", t), sd[ru + ".5.weight"], padding=1) + out
            shorts.append(out)
        out = F.conv2d(bn(mb + ".gate_unit.0", torch.cat(shorts + longs, 1)), sd[mb + ".gate_unit.2.weight"])
        longs.append(out)
    return F.conv2d(bn("reconstructor.0", out), sd["reconstructor.2.weight"]) + xi


def memnet_init_state_dict(num_memory_blocks: int, num_residual_blocks: int, in_chans: int = 1, seed: int = 0) -> SD:
    """Seeded weights in the shapes / key order of the reference net (network_memnet.py:100-116,172-179: Kaiming-normal
    convs), with BatchNorm weights, biases and running statistics moved off their initial 1 / 0 / 0 / 1 so that every
    one of them matters in the parity fixtures."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def bn(pre, c):
        sd[pre + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[pre + ".bias"] = 0.05 * torch.randn(c, generator=g)
        sd[pre + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[pre + ".running_var"] = 0.8 + 0.4 * torch.rand(c, generator=g)
        sd[pre + ".num_batches_tracked"] = torch.tensor(3, dtype=torch.long)

    def conv(name, co, ci, k):
        sd[name] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (ci * k * k))

    bn("feature_extractor.0", in_chans)
    conv("feature_extractor.2.weight", 64, in_chans, 3)
    for i in range(num_memory_blocks):
        mb = f"dense_memory_blocks.{i}"
        for j in range(num_residual_blocks):
            ru = f"{mb}.recursive_unit.{j}.residual_block"
            bn(ru + ".0", 64)
            conv(ru + ".2.weight", 64, 64, 3)
            bn(ru + ".3", 64)
            conv(ru + ".5.weight", 64, 64, 3)
        gc = (num_residual_blocks + i + 1) * 64
        bn(mb + ".gate_unit.0", gc)
        conv(mb + ".gate_unit.2.weight", 64, gc, 1)
    bn("reconstructor.0", 64)
    conv("reconstructor.2.weight", in_chans, 64, 1)
    return sd


# ----------------------------------------------------------------------------
# losses (dlib/loss/main.py, dlib/loss/ssim.py, dlib/loss/master.py)
# ----------------------------------------------------------------------------
def loss_l1(pred: Tensor, target: Tensor, lam: float = 1.0,
            weight: Optional[Tensor] = None) -> Tensor:
    """loss/main.py:45-76."""
    e = (pred - target).abs()
    if weight is not None:
        e = e * weight
    return lam * e.mean()


def loss_l2(pred: Tensor, target: Tensor, lam: float = 1.0) -> Tensor:
    """loss/main.py:79-99."""
    return lam * ((pred - target) ** 2).mean()


def gaussian_window_1d(ws: int, sigma: float = 1.5) -> Tensor:
    """loss/ssim.py:23-27 (float32 taps, normalised)."""
    g = torch.tensor([math.exp(-(i - ws // 2) ** 2 / float(2 * sigma ** 2))
                      for i in range(ws)], dtype=torch.float32)
    return g / g.sum()


def ssim_map_same(a: Tensor, b: Tensor, ws: int) -> Tensor:
    """loss/ssim.py:38-56: zero-padded 'same' depthwise Gaussian, C1=1e-4,
    C2=9e-4."""
    ch = a.shape[1]
    g = gaussian_window_1d(ws)
    win = (g[:, None] @ g[None, :]).float()[None, None].expand(ch, 1, ws, ws)
    win = win.contiguous().to(a)
    pad = ws // 2

    def blur(t):
        return F.conv2d(t, win, padding=pad, groups=ch)
    mu1, mu2 = blur(a), blur(b)
    s11 = blur(a * a) - mu1 * mu1
    s22 = blur(b * b) - mu2 * mu2
    s12 = blur(a * b) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / (
        (mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))


def loss_neg_ssim(pred: Tensor, target: Tensor, lam: float = 1.0,
                  ws: int = 11) -> Tensor:
    """loss/main.py:154-186: -lambda * mean_b(mean_chw(ssim_map))."""
    m = ssim_map_same(pred, target, ws)
    return -lam * m.mean(1).mean(1).mean(1).mean()


def loss_charbonnier(pred: Tensor, target: Tensor, lam: float = 1.0, eps: float = 1e-9) -> Tensor:
    """loss/main.py:125-151."""
    d = target - pred
    return lam * torch.sqrt(d * d + eps).mean()


def loss_l2sum(pred: Tensor, target: Tensor, lam: float = 1.0) -> Tensor:
    """loss/main.py:102-122: MSELoss(reduction='sum') (the trailing .mean() acts on a scalar)."""
    return lam * ((pred - target) ** 2).sum()


def local_variation_op(x: Tensor, kind: str, ksz: int = 3) -> Tensor:
    """loss/local_variations.py:18-141 on a 1-channel image, replicate padding.
    'grad': (x(.,+1) - x(.,-1), x(+1,.) - x(-1,.)); 'laplace': 8 x - the 8 neighbours;
    'lv': x - x(+off) for every off != 0 of a ksz x ksz window (row-major)."""
    assert x.ndim == 4 and x.shape[1] == 1
    h, w = x.shape[-2:]
    r = 1 if kind != "lv" else ksz // 2
    xp = F.pad(x, (r, r, r, r), mode="replicate")

    def sh(dy, dx):
        return xp[:, :, r + dy:r + dy + h, r + dx:r + dx + w]
    if kind == "grad":
        return torch.cat([sh(0, 1) - sh(0, -1), sh(1, 0) - sh(-1, 0)], 1)
    if kind == "laplace":
        nb = sum(sh(dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0))
        return 8 * x - nb
    if kind == "lv":
        return torch.cat([x - sh(i - r, j - r) for i in range(ksz) for j in range(ksz)
                          if (i, j) != (r, r)], 1)
    raise ValueError(kind)


def loss_local_variation(pred: Tensor, target: Tensor, kind: str, lam: float = 1.0, norm: int = 2,
                         ksz: int = 3, channel_norm: bool = False) -> Tensor:
    """loss/main.py:328-497 (ImageGradientLoss / LaplacianFilterLoss / LocalVariationLoss) and
    :500-674 (Norm*: 2-norm over the operator's channels first).  norm 1 = L1Loss, 2 = MSELoss,
    reduction 'none' then .mean()."""
    a, b = local_variation_op(pred, kind, ksz), local_variation_op(target, kind, ksz)
    if channel_norm:
        a, b = a.norm(p=2, dim=1, keepdim=True), b.norm(p=2, dim=1, keepdim=True)
    e = a - b
    return lam * (e.abs() if norm == 1 else e * e).mean()


def elb(z: Tensor, t: float) -> Tensor:
    """Extended log barrier, mean over the vector (dlib/losses/elb.py:92-122); ``t`` as the
    reference's float32 buffer ``t_lb``."""
    t = torch.tensor(t, dtype=torch.float32)
    ct = -(1. / (t ** 2))
    less = z <= ct
    zs = torch.where(less, z, torch.full_like(z, -1.0))      # keep log() off the other branch
    v_less = -(1. / t) * torch.log(-zs)
    v_great = t * z - (1. / t) * torch.log(1. / (t ** 2)) + (1. / t)
    return torch.where(less, v_less, v_great).mean()


def elb_t_after(updates: int, init_t: float = 1., max_t: float = 10., mulcoef: float = 1.01) -> float:
    """t after ``updates`` calls of ELB.update_t (elb.py:85-90), in float32."""
    t = torch.tensor([init_t], dtype=torch.float32)
    for _ in range(updates):
        t = torch.min(t * torch.tensor([mulcoef], dtype=torch.float32), torch.tensor([max_t], dtype=torch.float32))
    return float(t)


def loss_bounded_prediction(pred: Tensor, target: Tensor, lam: float = 1.0, eps: float = 0.0, t: float = 1.0,
                            restore_range: bool = False, color_max: int = 255) -> Tensor:
    """loss/main.py:189-237: y - eps <= y_hat <= y + eps through two ELB terms."""
    yh, y = pred.reshape(-1), target.reshape(-1)
    if restore_range:
        yh, y = yh * color_max, y * color_max
    return lam * (elb(yh - (y + eps), t) + elb(y - eps - yh, t)) / 2.


def loss_weights_sparsity(params: Sequence[Tensor], lam: float = 1.0) -> Tensor:
    """loss/main.py:938-959: lambda * sum_w ||w||_1."""
    return lam * sum(p.abs().sum() for p in params)


def patch_moments(x: Tensor, ksz: int = 3) -> Tuple[Tensor, Tensor]:
    """loss/local_terms.py:34-57: mean and unbiased variance of every ksz x ksz patch (reflect
    padding), [b, h*w] each."""
    p = (ksz - 1) // 2
    z = F.unfold(F.pad(x, (p, p, p, p), mode="reflect"), kernel_size=ksz).permute(0, 2, 1)
    var, mean = torch.var_mean(z, dim=-1, unbiased=True)
    return mean, var


def loss_local_moments(pred: Tensor, target: Tensor, lam: float = 1.0) -> Tensor:
    """loss/main.py:240-325.  The operators are built once for ksz = [3] (set_ksz does not rebuild
    them, :244-263), eps = 1; only patches whose TARGET variance is exactly 0 count."""
    sm, sv = patch_moments(pred)
    tm, tv = patch_moments(target)
    flat = (tv == 0).float()
    sv1, tv1 = sv + 1.0, tv + 1.0
    kl = torch.log(torch.sqrt(sv1) / torch.sqrt(tv1)) + (tv1 + (tm - sm) ** 2) / (2 * sv1) - 0.5
    return lam * (kl * flat).mean()


def soft_histogram(x: Tensor, bins: int = 256, sigma: float = 1e5) -> Tensor:
    """loss/global_terms.py:17-72 with min 0, max 1: x [b, n] -> [b, bins]."""
    delta = float(1.0) / float(bins)
    centers = 0.0 + delta * (torch.arange(bins).float() + 0.5)
    d = x.unsqueeze(1) - centers.unsqueeze(1)
    return (torch.sigmoid(sigma * (d + delta / 2.)) - torch.sigmoid(sigma * (d - delta / 2.))).sum(dim=-1)


def loss_histogram_match(pred: Tensor, target: Tensor, lam: float = 1.0, norm: int = 2, sigma: float = 1e5,
                         bins: int = 256, elb_t: float = 1.0) -> Tensor:
    """loss/main.py:690-782: histograms + 1, normalised; norm 1 | 2: compared bin by bin; 3: KL (nn.KLDivLoss
    batchmean on log p, :727-729,771-773); 4: Bhattacharyya through the extended log barrier at elb_t (:775-777)."""
    b = target.shape[0]
    t = soft_histogram(target.contiguous().view(b, -1), bins, sigma) + 1.
    t = t / t.sum(dim=-1).view(-1, 1)
    p = soft_histogram(pred.contiguous().view(b, -1), bins, sigma) + 1.
    p = p / p.sum(dim=-1).view(-1, 1)
    if norm == 3:
        return lam * F.kl_div(p.log(), t, reduction="batchmean", log_target=False).mean()
    if norm == 4:
        return lam * elb(-torch.sqrt(p * t).sum(dim=1).view(-1), elb_t)
    e = p - t
    return lam * (e.abs() if norm == 1 else e * e).mean()


def gaussian_kde(images: Tensor, kde_bw: float, bins: int = 256) -> Tensor:
    """loss/global_terms.py:75-152 for 1-channel images in [0, 1]: [b, 1, h, w] -> [b, bins]."""
    c1 = torch.tensor((2. * math.pi * kde_bw) ** (-1 / 2.), dtype=torch.float32)
    c2 = torch.tensor(2. * kde_bw, dtype=torch.float32)
    cs = torch.linspace(0., 1., bins, dtype=torch.float32).view(-1, 1)
    out = []
    for img in images:
        x = img.contiguous().view(1, 1, -1)
        out.append((c1 * torch.exp(-((x - cs) ** 2) / c2)).mean(dim=-1).squeeze(0))
    return torch.stack(out)


def loss_kde_match(pred: Tensor, target: Tensor, lam: float = 1.0, norm: int = 2, kde_bw: float = 1. / 255. ** 2,
                   bins: int = 256, elb_t: float = 1.0) -> Tensor:
    """loss/main.py:785-898: (kde + 1e-4); norm 1 | 2: compared bin by bin, mean / bins; 4: Bhattacharyya through the
    extended log barrier at elb_t (:891-892)."""
    t = gaussian_kde(target, kde_bw, bins) + 1e-4
    p = gaussian_kde(pred, kde_bw, bins) + 1e-4
    if norm == 4:
        return lam * elb(-torch.sqrt(p * t).sum(dim=1).view(-1), elb_t)
    e = p - t
    return lam * (e.abs() if norm == 1 else e * e).mean() / float(bins)


def master_loss(pred: Tensor, target: Tensor, terms: Sequence[tuple],
                weight: Optional[Tensor] = None) -> Tuple[Tensor, List[Tensor]]:
    """loss/master.py:46-56.  ``terms``: ('l1',lam) | ('l2',lam) |
    ('ssim',lam,ws) | ('charbonnier',lam,eps) | ('l2sum',lam) | ('grad'|'laplace'|'lv'|
    'norm_grad'|'norm_laplace'|'norm_lv', lam, norm[, ksz]).  Returns (total, l_holder) with
    l_holder[0] == total."""
    parts = []
    for t in terms:
        if t[0] == "l1":
            parts.append(loss_l1(pred, target, t[1], weight))
        elif t[0] == "l2":
            parts.append(loss_l2(pred, target, t[1]))
        elif t[0] == "ssim":
            parts.append(loss_neg_ssim(pred, target, t[1], t[2]))
        elif t[0] == "charbonnier":
            parts.append(loss_charbonnier(pred, target, t[1], t[2]))
        elif t[0] == "l2sum":
            parts.append(loss_l2sum(pred, target, t[1]))
        elif t[0] == "local_moments":
            parts.append(loss_local_moments(pred, target, t[1]))
        elif t[0] == "kde":                # (kind, lam, norm, kde_bw, bins)
            parts.append(loss_kde_match(pred, target, t[1], t[2], t[3], t[4]))
        elif t[0] == "hist":               # (kind, lam, norm, sigma, bins)
            parts.append(loss_histogram_match(pred, target, t[1], t[2], t[3], t[4]))
        elif t[0] == "boundpred":          # (kind, lam, eps, t, restore_range, color_max)
            parts.append(loss_bounded_prediction(pred, target, t[1], t[2], t[3], t[4], t[5]))
        elif t[0] in ("grad", "laplace", "lv", "norm_grad", "norm_laplace", "norm_lv"):
            # (kind, lam, norm[, ksz])
            parts.append(loss_local_variation(pred, target, t[0].replace("norm_", ""), t[1], t[2],
                                              t[3] if len(t) > 3 else 3, t[0].startswith("norm_")))
        else:
            raise ValueError(t[0])
    total = sum(parts)
    return total, [total] + parts


def roi_origin_pmf(img_u8, threshold: int, psize: int):
    """PatchSampler._roi (dataset_dpsr.py:330-347): probability of every candidate origin, (H-P) x (W-P):
    proportional to exp(5 * roi) + 1 with roi = img[r + P//2][c + P//2] >= threshold."""
    import numpy as np
    h, w = img_u8.shape
    lo, hi = int(psize / 2), math.ceil(psize / 2)
    roi = (img_u8 >= threshold).astype(np.float64)[lo:h - hi, lo:w - hi]
    tmp = np.exp(roi * 5.)
    return ((tmp.flatten() + 1.) / (tmp + 1.).sum()).reshape(roi.shape)


def roi_origin_from_uniform(img_u8, threshold: int, psize: int, u: float):
    """Inverse CDF of roi_origin_pmf in row-major order: the first origin whose cumulative WEIGHT
    W1 * (#roi origins so far) + W0 * (#others so far) exceeds u * (total weight), W1 = e^5 + 1, W0 = 2
    (counts are exact integers, so the device kernel can form the same fp64 numbers)."""
    import numpy as np
    h, w = img_u8.shape
    lo, hi = int(psize / 2), math.ceil(psize / 2)
    roi = (img_u8 >= threshold)[lo:h - hi, lo:w - hi].reshape(-1)
    w1, w0 = 149.4131591025766, 2.0
    assert w1 == math.exp(5.) + 1.
    c1 = np.cumsum(roi.astype(np.int64))
    c0 = np.arange(1, roi.size + 1, dtype=np.int64) - c1
    cum = w1 * c1.astype(np.float64) + w0 * c0.astype(np.float64)
    i = int(np.searchsorted(cum, u * cum[-1], side="right"))
    i = min(i, roi.size - 1)
    return i // (w - psize), i % (w - psize)


def augment_index(mode: int, i, j, P: int):
    """Source position inside a P x P patch of output position (i, j) under augment_img mode
    0..7 (utils_image.py:469-487: rot90 = counter-clockwise on axes (0, 1), flipud = rows reversed)."""
    e = P - 1
    return [(i, j), (j, i), (e - i, j), (e - j, i), (i, e - j), (j, e - i), (e - i, e - j), (e - j, e - i)][mode]


def patch_batch(tiles, ids, y0, x0, modes, P: int) -> Tensor:
    """The per-sample tail of DatasetDPSR.__getitem__ in training mode
    (dataset_dpsr.py:866-894,914-915): crop P x P at (y0, x0) from the uint8 tile, augment_img(mode),
    uint2single (np.float32(v / 255.), the division in float64), single2tensor3 -> [B, 1, P, P]."""
    import numpy as np
    out = np.empty((len(ids), 1, P, P), dtype=np.float32)
    ii, jj = np.meshgrid(np.arange(P), np.arange(P), indexing="ij")
    for b, (t, yy, xx, m) in enumerate(zip(ids, y0, x0, modes)):
        crop = np.asarray(tiles[t])[yy:yy + P, xx:xx + P]
        si, sj = augment_index(int(m), ii, jj, P)
        out[b, 0] = np.float32(crop[si, sj] / 255.)
    return torch.from_numpy(out)


def interpolate_bicubic(x: Tensor, scale: int) -> Tensor:
    """The Bicubic baseline (dlib/utils/utils_trainer.py:121-148): aten's antialiased bicubic
    resize, clamped to [0, 1]."""
    return torch.clamp(F.interpolate(x, scale_factor=scale, mode="bicubic", antialias=True), 0.0, 1.0)


# ----------------------------------------------------------------------------
# metrics (dlib/utils/utils_image.py)
# ----------------------------------------------------------------------------
def tensor2uint82float(img: Tensor) -> Tensor:
    """utils_image.py:369-372 (round half to even, as torch.round)."""
    return (img.float().clamp(0, 1) * 255.0).round().clamp(0, 255).float()


def _crop(t: Optional[Tensor], border: int) -> Optional[Tensor]:
    if t is None:
        return None
    h, w = t.shape[2:]
    return t[:, :, border:h - border, border:w - border]


def metric_mse(a: Tensor, b: Tensor, border: int = 0,
               roi: Optional[Tensor] = None) -> Tensor:
    """utils_image.py:894-934.  NB the ROI branch subtracts in the *input*
    dtype before promoting (reference behaviour)."""
    a, b, roi = _crop(a, border), _crop(b, border), _crop(roi, border)
    n = a.shape[0]
    if roi is None:
        return ((a.double() - b.double()) ** 2).reshape(n, -1).mean(-1)
    roi = roi.double()
    d = (a - b) * roi
    cnt = roi.reshape(n, -1).sum(-1)
    cnt[cnt == 0] = 1.0
    return (d ** 2).reshape(n, -1).sum(-1) / cnt


def metric_psnr(a: Tensor, b: Tensor, border: int = 0,
                roi: Optional[Tensor] = None) -> Tensor:
    """utils_image.py:843-891 (fp64; mse floor 1e-45)."""
    a, b, roi = _crop(a, border), _crop(b, border), _crop(roi, border)
    n = a.shape[0]
    a, b = a.double(), b.double()
    if roi is None:
        mse = ((a - b) ** 2).reshape(n, -1).mean(-1)
    else:
        roi = roi.double()
        cnt = roi.reshape(n, -1).sum(-1)
        cnt[cnt == 0] = 1.0
        mse = (((a - b) * roi) ** 2).reshape(n, -1).sum(-1) / cnt
    mse = torch.where(mse < 1e-45, torch.full_like(mse, 1e-45), mse)
    return 20.0 * torch.log10(255.0 / torch.sqrt(mse))


def metric_nrmse(img: Tensor, y: Tensor, border: int = 0,
                 roi: Optional[Tensor] = None) -> Tensor:
    """utils_image.py:937-1007."""
    img, y, roi = _crop(img, border), _crop(y, border), _crop(roi, border)
    n = img.shape[0]
    img, y = img.double(), y.double()
    if roi is None:
        mse = ((img - y) ** 2).reshape(n, -1).mean(-1)
        yy = y.reshape(n, -1)
        lo = yy.min(-1)[0]
    else:
        roi = roi.double()
        cnt = roi.reshape(n, -1).sum(-1)
        cnt[cnt == 0] = 1.0
        mse = (((img - y) * roi) ** 2).reshape(n, -1).sum(-1) / cnt
        lo_all = y.reshape(n, -1).min(-1)[0]
        yy = (y * roi).reshape(n, -1)
        lo = torch.maximum(lo_all, yy.min(-1)[0])
    den = yy.max(-1)[0] - lo
    den[den == 0] = 1.0
    return torch.sqrt(mse) / den


def gaussian_window_2d_metric(ks: int = 11, sigma: float = 1.5) -> Tensor:
    """utils_image.py:1102-1117."""
    c = torch.arange(ks, dtype=torch.float32) - (ks - 1) / 2.0
    g = (-(c[None, :] ** 2 + c[:, None] ** 2) / (2 * sigma ** 2)).exp()
    return g / g.sum()


def metric_ssim(x: Tensor, y: Tensor, border: int = 0,
                roi: Optional[Tensor] = None) -> Tensor:
    """utils_image.py:1120-1198 + :1010-1099: inputs in [0,255]; /255;
    11x11 sigma-1.5 'valid' Gaussian; per-image mean (ROI cropped by 5)."""
    x, y, roi = _crop(x, border), _crop(y, border), _crop(roi, border)
    x, y = x / 255.0, y / 255.0
    ch = x.shape[1]
    k = gaussian_window_2d_metric()[None, None].repeat(ch, 1, 1, 1).to(x)

    def blur(t):
        return F.conv2d(t, k, groups=ch)
    mx, my = blur(x), blur(y)
    sxx = blur(x ** 2) - mx ** 2
    syy = blur(y ** 2) - my ** 2
    sxy = blur(x * y) - mx * my
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    cs = (2.0 * sxy + c2) / (sxx + syy + c2)
    ss = ((2.0 * mx * my + c1) / (mx ** 2 + my ** 2 + c1)) * cs
    n = ss.shape[0]
    if roi is None:
        return ss.reshape(n, ch, -1).mean(-1).mean(1)
    r = roi[:, :, 5:roi.shape[2] - 5, 5:roi.shape[3] - 5]
    cnt = r.reshape(n, -1).sum(-1)
    cnt[cnt == 0] = 1.0
    return ((ss * r).reshape(n, ch, -1).sum(-1) / cnt[:, None]).mean(1)


def gray_to_y(img01: Tensor) -> Tensor:
    """_rgb_tensor (utils_trainer.py:865-871) + mb_gpu_rgb2ycbcr(only_y)
    (utils_image.py:618-653) for a 1-channel float image in [0,1]:
    repeat to 3ch, Y = ((65.481 r + 128.553 g + 24.966 b)/255 + 16)/255, clamp."""
    v = img01.float() * 255.0
    yv = (65.481 * v + 128.553 * v + 24.966 * v) / 255.0 + 16.0
    return (yv / 255.0).clamp(0.0, 1.0)


# ----------------------------------------------------------------------------
# optimizer / scheduler (utils_instance.py:216-247, learning/lr_scheduler.py:6-35)
# ----------------------------------------------------------------------------
def mysteplr(base_lr: float, it: int, step_size: int, gamma: float,
             min_lr: float) -> float:
    """LR in effect after ``it`` scheduler.step() calls."""
    return max(base_lr * gamma ** (it // step_size), min_lr)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1=0.9, b2=0.999, eps=1e-8, wd=0.0) -> None:
    """torch.optim.Adam (non-amsgrad, L2 weight decay), in place; ``step`` is
    1-based."""
    if wd:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)


def clip_grad_norm(grads, max_norm: float):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2) as the step calls it (model_plain.py:350-361,
    G_optimizer_clipgrad > 0): total = ||(||g_1||, ..., ||g_n||)||_2, coef = min(1, max_norm / (total + 1e-6)), every
    gradient scaled in place.  Returns (total, coef)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total, coef


def ema_update(e_params, g_params, decay: float) -> None:
    """ModelBase.update_E (model_base.py:213-219): e = e * decay + g * (1 - decay) per parameter, in place."""
    for e, g in zip(e_params, g_params):
        e.mul_(decay).add_(g, alpha=1 - decay)


def sgd_nesterov_step(p: Tensor, g: Tensor, buf: Tensor, first: bool, lr: float,
                      momentum=0.9, wd=0.0, nesterov=True) -> None:
    """torch.optim.SGD with momentum (dampening 0), in place."""
    if wd:
        g = g + wd * p
    if first:
        buf.copy_(g)
    else:
        buf.mul_(momentum).add_(g)
    p.add_(g + momentum * buf if nesterov else buf, alpha=-lr)
