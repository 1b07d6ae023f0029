"""CPU restatement of the FLUX AutoencoderKL (the two ends of the hot path).  TEST INFRASTRUCTURE.

Citations (D = /root/reference/diffusers/src/diffusers): AutoencoderKL.encode/decode
D/models/autoencoders/autoencoder_kl.py:263-332; Encoder / Decoder / DiagonalGaussianDistribution
D/models/autoencoders/vae.py:60-195, 198-360, 780-833; ResnetBlock2D D/models/resnet.py:320-373;
UNetMidBlock2D D/models/unets/unet_2d_blocks.py:589-741 with AttnProcessor2_0 D/models/attention_processor.py:2799-2881;
Downsample2D D/models/downsampling.py:132-150 (pad (0,1,0,1), stride 2); Upsample2D D/models/upsampling.py:142-192
(nearest x2 then conv).  Weights: flat dict with the reference's state-dict key names.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass(frozen=True)
class VaeConfig:
    in_channels: int = 3
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    latent_channels: int = 16
    norm_num_groups: int = 32
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159


def _conv(x, sd, name, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _gn(x, sd, name, groups):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)


def resnet(x: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    h = _conv(F.silu(_gn(x, sd, p + ".norm1", groups)), sd, p + ".conv1")
    h = _conv(F.silu(_gn(h, sd, p + ".norm2", groups)), sd, p + ".conv2")
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(x, sd, p + ".conv_shortcut", padding=0)
    return x + h


def mid_attention(x: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    B, C, H, W = x.shape
    res = x
    h = x.view(B, C, H * W).transpose(1, 2)
    h = F.group_norm(h.transpose(1, 2), groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps=1e-6).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]  # one head of dim C
    o = F.linear(o.to(q.dtype), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(-1, -2).reshape(B, C, H, W) + res


def mid_block(x, sd, p, groups):
    x = resnet(x, sd, p + ".resnets.0", groups)
    x = mid_attention(x, sd, p + ".attentions.0", groups)
    return resnet(x, sd, p + ".resnets.1", groups)


def encoder(x: Tensor, sd: SD, cfg: VaeConfig) -> Tensor:
    g = cfg.norm_num_groups
    h = _conv(x, sd, "encoder.conv_in")
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(h, sd, f"encoder.down_blocks.{i}.resnets.{j}", g)
        if i != n - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(h, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    h = mid_block(h, sd, "encoder.mid_block", g)
    return _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", g)), sd, "encoder.conv_out")


def decoder(z: Tensor, sd: SD, cfg: VaeConfig) -> Tensor:
    g = cfg.norm_num_groups
    h = _conv(z, sd, "decoder.conv_in")
    h = mid_block(h, sd, "decoder.mid_block", g)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(h, sd, f"decoder.up_blocks.{i}.resnets.{j}", g)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    return _conv(F.silu(_gn(h, sd, "decoder.conv_norm_out", g)), sd, "decoder.conv_out")


def encode_moments(x: Tensor, sd: SD, cfg: VaeConfig) -> Tuple[Tensor, Tensor]:
    """mean, std of the posterior (DiagonalGaussianDistribution.__init__, vae.py:781-793)."""
    mean, logvar = torch.chunk(encoder(x, sd, cfg), 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean, torch.exp(0.5 * logvar)


def sample_posterior(mean: Tensor, std: Tensor, eps: Tensor) -> Tensor:
    """DiagonalGaussianDistribution.sample (vae.py:795-802) with the drawn noise given explicitly."""
    return mean + std * eps


# --------------------------------------------------------------------------- tiled encode / decode (enable_tiling)
def _blend(a: Tensor, b: Tensor, extent: int, dim: int) -> Tensor:
    """AutoencoderKL.blend_v (dim 2) / blend_h (dim 3) (autoencoder_kl.py:334-344): the first `extent` rows / columns of b fade in from
    the last `extent` of a, IN PLACE on b (a tile that is blended later sees its already-blended neighbours, as in the reference);
    extent is clamped to both tiles and the weights use the clamped value.  The reference walks the seam line by line with python-float
    weights t / extent and 1 - t / extent; restated for the whole seam at once with the same arithmetic per element: the weights as
    fp32 (what a python float becomes beside a tensor), each product rounded to the tensor dtype, then the sum."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    if extent <= 0:
        return b
    t = torch.arange(extent, dtype=torch.float64) / extent
    shape = [1, 1, 1, 1]
    shape[dim] = extent
    wb, wa = t.to(torch.float32).view(shape), (1.0 - t).to(torch.float32).view(shape)
    tail, head = a.narrow(dim, a.shape[dim] - extent, extent), b.narrow(dim, 0, extent)
    mixed = (tail.float() * wa).to(b.dtype).float() + (head.float() * wb).to(b.dtype).float()
    head.copy_(mixed.to(b.dtype))
    return b


def _blend_v(a: Tensor, b: Tensor, extent: int) -> Tensor:
    return _blend(a, b, extent, 2)


def _blend_h(a: Tensor, b: Tensor, extent: int) -> Tensor:
    return _blend(a, b, extent, 3)


def _tiled(x: Tensor, fn, tile: int, out_tile: int, overlap_factor: float) -> Tensor:
    """The common body of _tiled_encode / tiled_decode (autoencoder_kl.py:346-395, 456-503): tiles of `tile` every
    int(tile * (1 - overlap)) pixels, each through `fn`; then, in raster order, every tile is blended over int(out_tile * overlap)
    output pixels with the tile above and the tile to its left (both already blended), cropped to out_tile - blend; the crops are
    concatenated."""
    step, extent = int(tile * (1 - overlap_factor)), int(out_tile * overlap_factor)
    keep = out_tile - extent
    ys, xs = range(0, x.shape[2], step), range(0, x.shape[3], step)
    grid = {(r, c): fn(x[:, :, y0:y0 + tile, x0:x0 + tile]) for r, y0 in enumerate(ys) for c, x0 in enumerate(xs)}
    bands = []
    for r in range(len(ys)):
        band = []
        for c in range(len(xs)):
            t = grid[r, c]
            if r:
                _blend(grid[r - 1, c], t, extent, 2)
            if c:
                _blend(grid[r, c - 1], t, extent, 3)
            band.append(t[:, :, :keep, :keep])
        bands.append(torch.cat(band, dim=3))
    return torch.cat(bands, dim=2)


def tile_sizes(cfg: VaeConfig, sample_size: int) -> Tuple[int, int]:
    """(tile_sample_min_size, tile_latent_min_size) of AutoencoderKL.__init__ (autoencoder_kl.py:131-138)."""
    return sample_size, int(sample_size / (2 ** (len(cfg.block_out_channels) - 1)))


def tiled_encoder(x: Tensor, sd: SD, cfg: VaeConfig, sample_size: int, overlap_factor: float = 0.25) -> Tensor:
    """AutoencoderKL._tiled_encode (autoencoder_kl.py:346-395; no quant conv in the FLUX VAE): the moments tensor.  _encode takes
    this path when use_tiling and the image is larger than tile_sample_min_size in either direction (:264-267)."""
    ts, tl = tile_sizes(cfg, sample_size)
    return _tiled(x, lambda t: encoder(t, sd, cfg), ts, tl, overlap_factor)


def tiled_decoder(z: Tensor, sd: SD, cfg: VaeConfig, sample_size: int, overlap_factor: float = 0.25) -> Tensor:
    """AutoencoderKL.tiled_decode (autoencoder_kl.py:456-503)."""
    ts, tl = tile_sizes(cfg, sample_size)
    return _tiled(z, lambda t: decoder(t, sd, cfg), tl, ts, overlap_factor)


def state_dict_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(name, o, i, k=3):
        out[name + ".weight"] = (o, i, k, k)
        out[name + ".bias"] = (o,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def res(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1", o, i); norm(p + ".norm2", o); conv(p + ".conv2", o, o)
        if i != o:
            conv(p + ".conv_shortcut", o, i, 1)

    def mid(p, c):
        for w in ("group_norm",):
            norm(f"{p}.attentions.0.{w}", c)
        for w in ("to_q", "to_k", "to_v", "to_out.0"):
            out[f"{p}.attentions.0.{w}.weight"] = (c, c)
            out[f"{p}.attentions.0.{w}.bias"] = (c,)
        res(p + ".resnets.0", c, c); res(p + ".resnets.1", c, c)

    boc = cfg.block_out_channels
    conv("encoder.conv_in", boc[0], cfg.in_channels)
    oc = boc[0]
    for i, c in enumerate(boc):
        ic, oc = oc, c
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", oc, oc)
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1])
    conv("encoder.conv_out", 2 * cfg.latent_channels, boc[-1])
    rev = list(reversed(boc))
    conv("decoder.conv_in", rev[0], cfg.latent_channels)
    oc = rev[0]
    for i, c in enumerate(rev):
        ic, oc = oc, c
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", oc, oc)
    mid("decoder.mid_block", rev[0])
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", cfg.out_channels, boc[0])
    return out


def seeded_state_dict(cfg: VaeConfig, seed: int = 0, dtype=torch.float32) -> SD:
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for k, shape in sorted(state_dict_shapes(cfg).items()):
        r = torch.randn(shape, generator=g)
        if "norm" in k and k.endswith(".weight"):
            t = 1.0 + 0.1 * r
        elif k.endswith(".bias"):
            t = 0.05 * r
        else:
            fan_in = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
            t = r / (fan_in ** 0.5)
        sd[k] = t.to(dtype)
    return sd
