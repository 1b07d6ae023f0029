"""FLUX AutoencoderKL for the two ends of the path (masked-image encode, final decode) on the gfx950 kernels.

Reference: AutoencoderKL.encode/decode D/models/autoencoders/autoencoder_kl.py:263-332 and the blocks cited in
oracle/vae_oracle.py.  Weights keep the reference's state-dict keys (the HF `vae/diffusion_pytorch_model.safetensors`
loads as is).  ONE execution path: activations are NHWC bf16 from the first to the last kernel;
* every 3x3 convolution is an implicit GEMM on the MFMA kernel (`tfx_conv3x3_nhwc`: the nearest-2x upsample and the
  (0,1,0,1)-padded stride-2 downsample are folded into its gather, the residual add into its epilogue); the two `conv_in`
  layers (3 / 16 input channels) use its narrow-input form (channels padded to 8 / 16, K padded to 128 / 192);
* GroupNorm(+SiLU) is `tfx_groupnorm_nhwc`; 1x1 shortcuts and the mid-block's q/k/v/out projections are `tfx_gemm_bf16`;
* the mid-block attention (ONE head of dim 512 over h*w tokens, AttnProcessor2_0 D/models/attention_processor.py:2799-2881)
  runs as fp32 scores = q k^T (`tfx_gemm_bf16_f32`) -> `tfx_row_softmax` (fp32 in, bf16 weights out) -> P v (MFMA GEMM
  against v^T from `tfx_transpose`), one image at a time through reused [N, N] score / weight buffers;
* the pipeline hands over / takes back NHWC tensors through `encode_moments_nhwc` / `decode_nhwc` (imageops.hip kernels on
  either side); `encode` / `decode` keep the reference's NCHW tensor interface on top of the same path.
Configurations the kernels do not cover (block widths that are not multiples of 64) raise at load time: there is no
torch / MIOpen fallback.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch

from . import ops


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)


class DiagonalGaussianDistribution:
    """mean / logvar split, logvar clamped to [-30, 20] (D/models/autoencoders/vae.py:781-802).  Interface object of the
    NCHW `encode()`; the pipeline itself samples + packs in one kernel (`tfx_vae_sample_pack`)."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from .pipeline import randn_tensor
        eps = randn_tensor(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


def _narrow(cin: int) -> int:
    """Channel count a narrow (< 64) conv input is padded to."""
    for c in (8, 16, 32):
        if cin <= c:
            return c
    raise ValueError(f"conv input width {cin} is not supported by the gfx950 conv kernel (<= 32 or a multiple of 64)")


class AutoencoderKL:
    def __init__(self, in_channels: int = 3, out_channels: int = 3, block_out_channels: Tuple[int, ...] = (128, 256, 512, 512),
                 layers_per_block: int = 2, latent_channels: int = 16, norm_num_groups: int = 32,
                 scaling_factor: float = 0.3611, shift_factor: float = 0.1159, sample_size: int = 32, **_ignored):
        self.config = _Config(in_channels=in_channels, out_channels=out_channels,
                              block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                              latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                              scaling_factor=scaling_factor, shift_factor=shift_factor, sample_size=sample_size)
        # tiling geometry (D/models/autoencoders/autoencoder_kl.py:127-139; the FLUX VAE's config.json says sample_size 1024)
        ss = sample_size[0] if isinstance(sample_size, (list, tuple)) else sample_size
        self.use_tiling = False
        self.tile_sample_min_size = ss
        self.tile_latent_min_size = int(ss / (2 ** (len(self.config.block_out_channels) - 1)))
        self.tile_overlap_factor = 0.25
        bad = [ch for ch in self.config.block_out_channels
               if ch % 64 or (ch // norm_num_groups) % 4 or 256 % (ch // 8) or ch % norm_num_groups]
        if bad or in_channels > 8 or out_channels > 8 or latent_channels % 8 or latent_channels > 32:
            raise ValueError(f"AutoencoderKL config outside the gfx950 kernels' range: block widths {bad or '-'} must be "
                             "multiples of 64 with (width / groups) % 4 == 0; in/out channels <= 8; latent channels 8/16/24/32")
        self.pair_convs = True     # 3 x 3 layers with <= 128 output channels in pixel-pair form (tfx_conv3x3_pair_nhwc); A/B knob: set before load
        self.sd: Dict[str, torch.Tensor] = {}
        self.hw: Dict[str, torch.Tensor] = {}
        self.dtype, self.device = torch.bfloat16, torch.device("cpu")
        self._scores = None     # (fp32 scores [N, Np], bf16 softmax weights [N, Np]) of the mid-block attention, reused

    # ---- weights ------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], device="cuda", dtype=torch.bfloat16):
        if dtype != torch.bfloat16:
            raise ValueError("the HIP VAE computes in bf16")
        self.sd = {k: v.to(device=device, dtype=dtype) for k, v in sd.items()}
        self.dtype, self.device = dtype, torch.device(device)
        self._prep()
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
                "norm_num_groups", "scaling_factor", "shift_factor", "sample_size")
        m = cls(**{k: cfg[k] for k in keys if k in cfg})
        sd = load_file(os.path.join(root, "diffusion_pytorch_model.safetensors"))
        missing = [k for k in m._shapes() if k not in sd]
        if missing:
            raise RuntimeError(f"VAE checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        return m.load_state_dict(sd, device, torch_dtype or torch.bfloat16)

    def to(self, device=None, dtype=None):
        if self.sd and (device is not None or dtype is not None):
            self.load_state_dict(self.sd, device or self.device, dtype or self.dtype)
        return self

    def _prep(self):
        """Kernel-side weight layouts: 3x3 filters as [Cout, 3, 3, Cin] (KRSC) rows -- narrow inputs channel-padded and
        K-padded with zeros, conv_out's 3 filters padded to 8 --, 1x1 shortcuts as matrices."""
        self.hw = {}
        for k, v in self.sd.items():
            if not (k.endswith(".weight") and v.dim() == 4):
                continue
            name = k[:-7]
            if v.shape[2] == 1:
                self.hw[k] = v.reshape(v.shape[0], v.shape[1]).contiguous()
                continue
            w = v.permute(0, 2, 3, 1).contiguous()                       # [Cout, 3, 3, Cin]
            cout, cin = w.shape[0], w.shape[3]
            if cin % 64:
                cp = _narrow(cin)
                wp = torch.zeros(cout, 3, 3, cp, dtype=w.dtype, device=w.device)
                wp[..., :cin] = w
                kp = (9 * cp + 63) // 64 * 64
                w = torch.zeros(cout, kp, dtype=w.dtype, device=w.device)
                w[:, :9 * cp] = wp.reshape(cout, 9 * cp)
            if cout % 8:
                pad = 8 - cout % 8
                w = torch.cat([w, torch.zeros(pad, *w.shape[1:], dtype=w.dtype, device=w.device)], 0)
                b = self.sd[name + ".bias"]
                self.hw[name + ".bias"] = torch.cat([b, torch.zeros(pad, dtype=b.dtype, device=b.device)], 0)
            self.hw[k] = w.contiguous()
            # few output channels (the full-resolution blocks: 128): the pixel-pair form fills the MFMA kernel's 256-column tile
            if self.pair_convs and cin % 64 == 0 and cout % 8 == 0 and cout <= 128:
                self.hw[name + ".pair"] = ops.pair_conv_weights(w.contiguous(), self.sd[name + ".bias"])

    # ---- building blocks ----------------------------------------------------------------------------------------
    def _conv(self, x, name, **kw):
        pair = self.hw.get(name + ".pair")
        if pair is not None and x.shape[2] % 2 == 0 and kw.get("stride", 1) == 1 and kw.get("up", 1) == 1 and kw.get("pad_lo", 1) == 1:
            return ops.conv3x3_pair_nhwc(x, pair[0], pair[1], res=kw.get("res"))
        return ops.conv3x3_nhwc(x, self.hw[name + ".weight"], self.hw.get(name + ".bias", self.sd[name + ".bias"]), **kw)

    def _gn(self, x, name, silu=True):
        return ops.groupnorm_nhwc(x, self.sd[name + ".weight"], self.sd[name + ".bias"], self.config.norm_num_groups, silu=silu)

    def _res(self, x, p):
        h = self._conv(self._gn(x, p + ".norm1"), p + ".conv1")
        h = self._gn(h, p + ".norm2")
        if p + ".conv_shortcut.weight" in self.sd:
            B, H, W, C = x.shape
            x = ops.gemm(x.view(B * H * W, C), self.hw[p + ".conv_shortcut.weight"], self.sd[p + ".conv_shortcut.bias"]).view(B, H, W, -1)
        return self._conv(h, p + ".conv2", res=x)

    # query rows per score chunk of the mid-block attention: the fp32 scores + bf16 weights of ONE chunk are the whole scratch
    # (<= ATTEND_CHUNK_BYTES whatever the image size: 1024 x 1024 -> N = 16384 keys, 2048 rows per chunk, 8 chunks per image)
    ATTEND_CHUNK_BYTES = 192 << 20

    def _attend(self, q, k, v):
        """softmax(q k^T / sqrt(C)) v for ONE head of dim C, q / k / v [B, N, C] bf16 (AttnProcessor2_0,
        D/models/attention_processor.py:2799-2881).  Scores stay fp32 between the two GEMMs (a flash kernel would keep them
        in registers); token counts are padded to a multiple of 64 with zero weights / zero v^T columns so that the P v
        product takes the MFMA kernel for any h * w.  Query rows are independent, so the score matrix is never materialised as a
        whole: the whole batch at a time (round 6), `rows` query rows at a time through ONE reused [B, rows, Np] score / weight buffer pair whose size
        does not grow with the image (round 4 held the full [N, N]: 1.5 GiB at 1024 x 1024, 6 GiB at 2048 x 1024).  Scores and
        softmax weights of a row do not depend on the chunking (bit-identical); the P v sum over the keys is K-sliced according to the
        chunk's tile count, so two chunk sizes agree to the last bf16 ulp, not bit for bit (tests/test_vae_kernels_gpu.py)."""
        B, N, C = q.shape
        Np = (N + 63) // 64 * 64
        vt = torch.zeros(B, C, Np, dtype=torch.bfloat16, device=q.device) if Np != N else torch.empty(B, C, N, dtype=torch.bfloat16, device=q.device)
        ops.transpose(v, out=vt[:, :, :N])
        # round 6: the whole BATCH per launch (k and v^T are per-image weight matrices: tfx_gemm_args.w_bstride) -- three launches per
        # query-row chunk instead of three per image and chunk (1024 x 1024, batch 8: 24 instead of 192 per attention); the chunk's rows
        # shrink with the batch so that the score / weight scratch keeps its bound
        rows = max(256, min(N, self.ATTEND_CHUNK_BYTES // (B * Np * 6) // 256 * 256))
        if self._scores is None or self._scores[0].shape != (B, rows, Np):
            self._scores = None                                          # release before re-allocating
            self._scores = (torch.empty(B, rows, Np, dtype=torch.float32, device=q.device),
                            torch.zeros(B, rows, Np, dtype=torch.bfloat16, device=q.device))
        s, pw = self._scores
        # the P v product of one chunk has B x rows / 256 x C / 256 tiles for 256 CUs and K = Np: where that is fewer than the CUs it runs
        # K-sliced through the GEMM's split-K scratch (fp32 partials, fixed order: deterministic for a given image and batch size)
        if getattr(self, "_pv_ws", None) is None or self._pv_ws.device != q.device:
            self._pv_ws = torch.empty(64 << 20, dtype=torch.uint8, device=q.device)
        o = torch.empty(B, N, C, dtype=torch.bfloat16, device=q.device)
        for r0 in range(0, N, rows):
            n = min(rows, N - r0)
            ops.gemm_f32(q[:, r0:r0 + n], k, out=s[:, :n, :N])
            if n == rows:                                   # rows of all images at one stride: one launch
                ops.row_softmax(s.view(B * rows, Np)[:, :N], C ** -0.5, pw.view(B * rows, Np))
            else:                                           # the ragged last chunk: one per image
                for b in range(B):
                    ops.row_softmax(s[b, :n, :N], C ** -0.5, pw[b, :n])
            ops.gemm(pw[:, :n], vt, None, out=o[:, r0:r0 + n], workspace=self._pv_ws)
        return o

    def _mid(self, x, p):
        x = self._res(x, p + ".resnets.0")
        B, H, W, C = x.shape
        N = H * W
        a = p + ".attentions.0"
        tok = x.view(B, N, C)
        h = self._gn(tok, a + ".group_norm", silu=False)
        q, k, v = (ops.gemm(h, self.sd[f"{a}.{n}.weight"], self.sd[f"{a}.{n}.bias"]) for n in ("to_q", "to_k", "to_v"))
        o = self._attend(q, k, v)
        x = ops.gemm(o, self.sd[a + ".to_out.0.weight"], self.sd[a + ".to_out.0.bias"], epilogue=ops.EPI_BIAS_RES,
                     res=tok).view(B, H, W, C)
        return self._res(x, p + ".resnets.1")

    # ---- NHWC entry points (what the pipeline uses) ----------------------------------------------------------------
    # enable_slicing() / disable_slicing(): the reference's memory knob (D/models/autoencoders/autoencoder_kl.py:120-132,
    # 263-275, 301-306): with it, encode / decode run one sample at a time and concatenate.  Here every row of an implicit-GEMM
    # convolution and every GroupNorm group depends on its own sample only, so the sliced result is bit-identical to the batched
    # one (tests/test_vae_kernels_gpu.py); it bounds the activation working set to one image's.
    use_slicing = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    # enable_tiling() / disable_tiling(): the reference's other memory knob (autoencoder_kl.py:145-160): inputs larger than
    # tile_sample_min_size (tile_latent_min_size for decode) in either direction are cut into overlapping tiles, each tile runs through
    # the whole encoder / decoder on its own (its own GroupNorm statistics, its own mid-block attention), neighbouring results are
    # blended over a quarter tile and cropped (:346-395, 456-503).  This computes a DIFFERENT image from the untiled path -- in the
    # reference too -- and is reproduced here tile for tile, blend for blend (tfx_blend_edge_nhwc), in the reference's order: a tile is
    # blended IN PLACE with its upper and left neighbours, which have already been blended with theirs.
    def enable_tiling(self, use_tiling: bool = True):
        self.use_tiling = use_tiling

    def disable_tiling(self):
        self.enable_tiling(False)

    def _tiled(self, x: torch.Tensor, fn, tile: int, out_tile: int) -> torch.Tensor:
        step = int(tile * (1 - self.tile_overlap_factor))
        extent = int(out_tile * self.tile_overlap_factor)
        limit = out_tile - extent
        rows = [[fn(x[:, i:i + tile, j:j + tile].contiguous()) for j in range(0, x.shape[2], step)] for i in range(0, x.shape[1], step)]
        for i, row in enumerate(rows):
            for j, t in enumerate(row):
                if i > 0:
                    ops.blend_edge_nhwc_(rows[i - 1][j], t, extent, axis=1)
                if j > 0:
                    ops.blend_edge_nhwc_(row[j - 1], t, extent, axis=2)
        hs = [min(r[0].shape[1], limit) for r in rows]
        ws = [min(t.shape[2], limit) for t in rows[0]]
        out = torch.empty(x.shape[0], sum(hs), sum(ws), rows[0][0].shape[3], dtype=rows[0][0].dtype, device=x.device)
        y0 = 0
        for i, row in enumerate(rows):
            x0 = 0
            for j, t in enumerate(row):
                out[:, y0:y0 + hs[i], x0:x0 + ws[j]].copy_(t[:, :hs[i], :ws[j]])       # crop + concatenate: device copies
                x0 += ws[j]
            y0 += hs[i]
        return out

    def _sliced(self, fn, x: torch.Tensor) -> torch.Tensor:
        if not self.use_slicing or x.shape[0] <= 1:
            return fn(x)
        out = None
        for i in range(x.shape[0]):
            o = fn(x[i:i + 1])
            if out is None:
                out = torch.empty(x.shape[0], *o.shape[1:], dtype=o.dtype, device=o.device)
            out[i:i + 1].copy_(o)        # device-to-device copy of one sample's result
        return out

    @torch.no_grad()
    def encode_moments_nhwc(self, x8: torch.Tensor) -> torch.Tensor:
        """x8 [B, H, W, 8] NHWC bf16 (tfx_prep_image) -> posterior moments [B, H/8, W/8, 2 * latent] NHWC (mean | logvar)."""
        return self._sliced(self._encode_moments_nhwc, x8)

    def _encode_moments_nhwc(self, x8: torch.Tensor) -> torch.Tensor:
        if self.use_tiling and (x8.shape[2] > self.tile_sample_min_size or x8.shape[1] > self.tile_sample_min_size):   # :264-267
            return self._tiled(x8, self._encode_moments_whole, self.tile_sample_min_size, self.tile_latent_min_size)
        return self._encode_moments_whole(x8)

    def _encode_moments_whole(self, x8: torch.Tensor) -> torch.Tensor:
        c = self.config
        h = self._conv(x8, "encoder.conv_in")
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block):
                h = self._res(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._conv(h, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, pad_lo=0)
        h = self._mid(h, "encoder.mid_block")
        return self._conv(self._gn(h, "encoder.conv_norm_out"), "encoder.conv_out")

    @torch.no_grad()
    def decode_nhwc(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, h, w, latent] NHWC bf16 -> image [B, 8h, 8w, 8] NHWC bf16 (first out_channels channels valid)."""
        return self._sliced(self._decode_nhwc, z)

    def _decode_nhwc(self, z: torch.Tensor) -> torch.Tensor:
        if self.use_tiling and (z.shape[2] > self.tile_latent_min_size or z.shape[1] > self.tile_latent_min_size):       # :301-303
            return self._tiled(z, self._decode_whole, self.tile_latent_min_size, self.tile_sample_min_size)
        return self._decode_whole(z)

    def _decode_whole(self, z: torch.Tensor) -> torch.Tensor:
        c = self.config
        if z.shape[-1] != _narrow(c.latent_channels):
            zp = torch.zeros(*z.shape[:-1], _narrow(c.latent_channels), dtype=z.dtype, device=z.device)
            zp[..., : z.shape[-1]] = z
            z = zp
        h = self._conv(z.contiguous(), "decoder.conv_in")
        h = self._mid(h, "decoder.mid_block")
        n = len(c.block_out_channels)
        for i in range(n):
            for j in range(c.layers_per_block + 1):
                h = self._res(h, f"decoder.up_blocks.{i}.resnets.{j}")
            if i != n - 1:
                h = self._conv(h, f"decoder.up_blocks.{i}.upsamplers.0.conv", up=2)
        return self._conv(self._gn(h, "decoder.conv_norm_out"), "decoder.conv_out")

    # ---- the reference's NCHW tensor interface -------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        x = x.to(self.device)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        mom = self.encode_moments_nhwc(ops.prep_image(x, None, norm_mode=0))
        B, h, w, C2 = mom.shape
        post = DiagonalGaussianDistribution(ops.transpose(mom.view(B, h * w, C2)).view(B, C2, h, w))
        return SimpleNamespace(latent_dist=post) if return_dict else (post,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        z = z.to(self.device, self.dtype).contiguous()
        B, L, h, w = z.shape
        out = self.decode_nhwc(ops.transpose(z.view(B, L, h * w)).view(B, h, w, L))
        out = ops.postprocess(out, self.config.out_channels, "pt", denorm=False)
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
