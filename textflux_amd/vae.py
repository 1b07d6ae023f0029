"""FLUX AutoencoderKL for the two ends of the path (masked-image encode, final decode).

Reference: AutoencoderKL.encode/decode D/models/autoencoders/autoencoder_kl.py:263-332 and the blocks cited in
oracle/vae_oracle.py.  Two execution paths over the same weights (reference state-dict keys, so the HF
`vae/diffusion_pytorch_model.safetensors` loads as is):
* HIP / NHWC (default whenever every block width is a multiple of 64, i.e. the FLUX VAE 128/256/512): activations stay
  NHWC bf16; every 3x3 convolution is an implicit GEMM on the MFMA kernel (`tfx_conv3x3_nhwc`: nearest-2x upsample and
  the (0,1,0,1)-padded stride-2 downsample are folded into its gather, the residual add into its epilogue), GroupNorm+SiLU
  is `tfx_groupnorm_nhwc`, 1x1 shortcuts and the mid-block attention projections are plain `tfx_gemm_bf16` calls.  Still
  through torch: conv_in (3 or 16 input channels), the single-head dim-512 mid-block softmax (F.scaled_dot_product_attention).
* torch / NCHW (tiny test configurations whose widths are not multiples of 64): F.conv2d / F.group_norm via MIOpen.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)


class DiagonalGaussianDistribution:
    """mean / logvar split, logvar clamped to [-30, 20] (D/models/autoencoders/vae.py:781-802)."""

    def __init__(self, parameters: torch.Tensor):
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from .pipeline import randn_tensor
        eps = randn_tensor(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


class AutoencoderKL:
    def __init__(self, in_channels: int = 3, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2, latent_channels: int = 16, norm_num_groups: int = 32,
                 scaling_factor: float = 0.3611, shift_factor: float = 0.1159, **_ignored):
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                              scaling_factor=scaling_factor, shift_factor=shift_factor)
        self.sd: Dict[str, torch.Tensor] = {}
        self.dtype, self.device = torch.bfloat16, torch.device("cpu")
        self.use_hip, self.hw = False, {}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device="cuda", dtype=torch.bfloat16):
        self.sd = {k: v.to(device=device, dtype=dtype) for k, v in sd.items()}
        self.dtype, self.device = dtype, torch.device(device)
        self._prep_hip()
        return self

    def init_random_(self, seed: int = 0, device="cuda", dtype=torch.bfloat16):
        """Synthetic weights of the right shapes (benchmarks; no checkpoints offline)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shape in self._shapes().items():
            r = torch.randn(shape, generator=g)
            if "norm" in k and k.endswith(".weight"):
                sd[k] = 1 + 0.1 * r
            elif k.endswith(".bias"):
                sd[k] = 0.02 * r
            else:
                fan = shape[1] * (shape[2] * shape[3] if len(shape) == 4 else 1)
                sd[k] = r / fan ** 0.5
        return self.load_state_dict(sd, device, dtype)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype=torch.bfloat16, device="cuda", **_):
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "config.json")) as f:
            cfg = json.load(f)
        keys = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "latent_channels",
                "norm_num_groups", "scaling_factor", "shift_factor")
        m = cls(**{k: cfg[k] for k in keys if k in cfg})
        return m.load_state_dict(load_file(os.path.join(root, "diffusion_pytorch_model.safetensors")), device, torch_dtype)

    def to(self, device=None, dtype=None):
        if self.sd and (device is not None or dtype is not None):
            self.load_state_dict(self.sd, device or self.device, dtype or self.dtype)
        return self

    # ---- HIP / NHWC path ---------------------------------------------------------------------------------------
    def _prep_hip(self):
        c = self.config
        cpg_ok = all((ch // c.norm_num_groups) % 4 == 0 and 256 % (ch // 8) == 0 for ch in c.block_out_channels)
        self.use_hip = (self.device.type == "cuda" and self.dtype == torch.bfloat16 and cpg_ok
                        and all(ch % 64 == 0 for ch in c.block_out_channels))
        self.hw: Dict[str, torch.Tensor] = {}
        if not self.use_hip:
            return
        for k, v in self.sd.items():
            if k.endswith(".weight") and v.dim() == 4:
                if v.shape[2] == 3 and v.shape[1] % 64 == 0:
                    w = v.permute(0, 2, 3, 1).contiguous()                   # [Cout, 3, 3, Cin]  (KRSC)
                    if w.shape[0] % 8:                                       # conv_out: pad Cout 3 -> 8 (zero filters)
                        pad = 8 - w.shape[0] % 8
                        w = torch.cat([w, torch.zeros(pad, *w.shape[1:], dtype=w.dtype, device=w.device)], 0)
                        b = self.sd[k[:-7] + ".bias"]
                        self.hw[k[:-7] + ".bias"] = torch.cat([b, torch.zeros(pad, dtype=b.dtype, device=b.device)], 0)
                    self.hw[k] = w
                elif v.shape[2] == 1:
                    self.hw[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()   # 1x1 shortcut as a matrix

    def _hconv(self, x, name, **kw):
        from . import ops
        return ops.conv3x3_nhwc(x, self.hw[name + ".weight"], self.hw.get(name + ".bias", self.sd[name + ".bias"]), **kw)

    def _hgn(self, x, name, silu=True):
        from . import ops
        return ops.groupnorm_nhwc(x, self.sd[name + ".weight"], self.sd[name + ".bias"], self.config.norm_num_groups, silu=silu)

    def _hres(self, x, p):
        from . import ops
        h = self._hconv(self._hgn(x, p + ".norm1"), p + ".conv1")
        h = self._hgn(h, p + ".norm2")
        if p + ".conv_shortcut.weight" in self.sd:
            B, H, W, C = x.shape
            x = ops.gemm(x.view(B * H * W, C), self.hw[p + ".conv_shortcut.weight"], self.sd[p + ".conv_shortcut.bias"]).view(B, H, W, -1)
        return self._hconv(h, p + ".conv2", res=x)

    def _hmid(self, x, p):
        from . import ops
        x = self._hres(x, p + ".resnets.0")
        B, H, W, C = x.shape
        a = p + ".attentions.0"
        tok = x.view(B, H * W, C)
        h = self._hgn(tok, a + ".group_norm", silu=False)
        q, k, v = (ops.gemm(h, self.sd[f"{a}.{n}.weight"], self.sd[f"{a}.{n}.bias"]) for n in ("to_q", "to_k", "to_v"))
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0].contiguous()   # one head of dim C
        x = ops.gemm(o, self.sd[a + ".to_out.0.weight"], self.sd[a + ".to_out.0.bias"], epilogue=ops.EPI_BIAS_RES,
                     res=tok).view(B, H, W, C)
        return self._hres(x, p + ".resnets.1")

    def _encoder_hip(self, x):
        c = self.config
        h = F.conv2d(x, self.sd["encoder.conv_in.weight"], self.sd["encoder.conv_in.bias"], padding=1)
        h = h.permute(0, 2, 3, 1).contiguous()
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block):
                h = self._hres(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._hconv(h, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, pad_lo=0)
        h = self._hmid(h, "encoder.mid_block")
        h = self._hconv(self._hgn(h, "encoder.conv_norm_out"), "encoder.conv_out")
        return h.permute(0, 3, 1, 2).contiguous()

    def _decoder_hip(self, z):
        c = self.config
        h = F.conv2d(z, self.sd["decoder.conv_in.weight"], self.sd["decoder.conv_in.bias"], padding=1)
        h = h.permute(0, 2, 3, 1).contiguous()
        h = self._hmid(h, "decoder.mid_block")
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block + 1):
                h = self._hres(h, f"decoder.up_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._hconv(h, f"decoder.up_blocks.{i}.upsamplers.0.conv", up=2)
        h = self._hconv(self._hgn(h, "decoder.conv_norm_out"), "decoder.conv_out")
        return h[..., : c.out_channels].permute(0, 3, 1, 2).contiguous()

    # ---- building blocks (torch ops on the ROCm device) ------------------------------------------------------
    def _conv(self, x, name, stride=1, padding=1):
        return F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=padding)

    def _gn(self, x, name):
        return F.group_norm(x, self.config.norm_num_groups, self.sd[name + ".weight"], self.sd[name + ".bias"], eps=1e-6)

    def _resnet(self, x, p):
        h = self._conv(F.silu(self._gn(x, p + ".norm1")), p + ".conv1")
        h = self._conv(F.silu(self._gn(h, p + ".norm2")), p + ".conv2")
        if p + ".conv_shortcut.weight" in self.sd:
            x = self._conv(x, p + ".conv_shortcut", padding=0)
        return x + h

    def _mid(self, x, p):
        x = self._resnet(x, p + ".resnets.0")
        B, C, H, W = x.shape
        a = p + ".attentions.0"
        h = self._gn(x.view(B, C, H * W), a + ".group_norm").transpose(1, 2)
        q, k, v = (F.linear(h, self.sd[f"{a}.{n}.weight"], self.sd[f"{a}.{n}.bias"]) for n in ("to_q", "to_k", "to_v"))
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = F.linear(o, self.sd[a + ".to_out.0.weight"], self.sd[a + ".to_out.0.bias"])
        x = o.transpose(-1, -2).reshape(B, C, H, W) + x
        return self._resnet(x, p + ".resnets.1")

    def _encoder(self, x):
        c = self.config
        h = self._conv(x, "encoder.conv_in")
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block):
                h = self._resnet(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._conv(F.pad(h, (0, 1, 0, 1)), f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
        h = self._mid(h, "encoder.mid_block")
        return self._conv(F.silu(self._gn(h, "encoder.conv_norm_out")), "encoder.conv_out")

    def _decoder(self, z):
        c = self.config
        h = self._conv(z, "decoder.conv_in")
        h = self._mid(h, "decoder.mid_block")
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block + 1):
                h = self._resnet(h, f"decoder.up_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), f"decoder.up_blocks.{i}.upsamplers.0.conv")
        return self._conv(F.silu(self._gn(h, "decoder.conv_norm_out")), "decoder.conv_out")

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        x = x.to(self.device, self.dtype)
        post = DiagonalGaussianDistribution(self._encoder_hip(x) if self.use_hip else self._encoder(x))
        return SimpleNamespace(latent_dist=post) if return_dict else (post,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        z = z.to(self.device, self.dtype)
        out = self._decoder_hip(z) if self.use_hip else self._decoder(z)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def _shapes(self):
        c = self.config
        out = {}

        def conv(n, o, i, k=3):
            out[n + ".weight"], out[n + ".bias"] = (o, i, k, k), (o,)

        def norm(n, ch):
            out[n + ".weight"], out[n + ".bias"] = (ch,), (ch,)

        def res(p, i, o):
            norm(p + ".norm1", i); conv(p + ".conv1", o, i); norm(p + ".norm2", o); conv(p + ".conv2", o, o)
            if i != o:
                conv(p + ".conv_shortcut", o, i, 1)

        def mid(p, ch):
            norm(p + ".attentions.0.group_norm", ch)
            for w in ("to_q", "to_k", "to_v", "to_out.0"):
                out[f"{p}.attentions.0.{w}.weight"], out[f"{p}.attentions.0.{w}.bias"] = (ch, ch), (ch,)
            res(p + ".resnets.0", ch, ch); res(p + ".resnets.1", ch, ch)

        boc = c.block_out_channels
        conv("encoder.conv_in", boc[0], c.in_channels)
        oc = boc[0]
        for i, ch in enumerate(boc):
            ic, oc = oc, ch
            for j in range(c.layers_per_block):
                res(f"encoder.down_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
            if i != len(boc) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", oc, oc)
        mid("encoder.mid_block", boc[-1]); norm("encoder.conv_norm_out", boc[-1])
        conv("encoder.conv_out", 2 * c.latent_channels, boc[-1])
        rev = list(reversed(boc))
        conv("decoder.conv_in", rev[0], c.latent_channels)
        oc = rev[0]
        for i, ch in enumerate(rev):
            ic, oc = oc, ch
            for j in range(c.layers_per_block + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", ic if j == 0 else oc, oc)
            if i != len(rev) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", oc, oc)
        mid("decoder.mid_block", rev[0]); norm("decoder.conv_norm_out", boc[0])
        conv("decoder.conv_out", c.out_channels, boc[0])
        return out
