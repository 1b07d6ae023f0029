"""Plain-PyTorch CPU restatement of the FLUX.1-Fill denoiser used by TextFlux.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: weights live in
a flat ``dict[str, Tensor]`` that uses the reference's own state-dict key names
(SURVEY.md Appendix A), so the same seeded weights can be loaded into the
imported reference module when goldens are generated.

The op order, and therefore every rounding point when run in bf16, follows the
reference line by line (citations are to /root/reference/diffusers/src/diffusers,
abbreviated ``D/``).  Run with fp32 tensors for the fp32 oracle, with bf16
tensors for the bf16-faithful oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- config
@dataclass(frozen=True)
class FluxConfig:
    """Mirrors the register_to_config kwargs of FluxTransformer2DModel
    (D/models/transformers/transformer_flux.py:866-879)."""

    patch_size: int = 1
    in_channels: int = 384
    out_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, ...] = (16, 56, 56)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# fp8 emulation (BASELINE config 5; the reference has no fp8 path -- this restates the BUILD's documented scheme so the
# fp8 engine can be checked the same way as the bf16 one): inside `with fp8_block_linears():` every nn.Linear of the
# 57 blocks (not the norm*.linear modulation layers, not the embedders / proj_out) quantises its input rows and its
# weight rows to OCP e4m3 with a per-row scale absmax / 448, multiplies in fp32 and rescales.
_FP8 = {"on": False}


class fp8_block_linears:
    def __enter__(self):
        _FP8["on"] = True

    def __exit__(self, *a):
        _FP8["on"] = False


def quantize_rows_e4m3(x: Tensor) -> Tuple[Tensor, Tensor]:
    """(q, scale): q = round_e4m3(x / scale) as fp32 values, scale = max|row| / 448 (1 for a zero row), fp32 [.., 1]."""
    xf = x.float()
    s = xf.abs().amax(-1, keepdim=True) / 448.0
    s = torch.where(s > 0, s, torch.ones_like(s))
    return (xf / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float(), s


def _is_block_linear(name: str) -> bool:
    return name.startswith(("transformer_blocks.", "single_transformer_blocks.")) and ".norm" not in name


def linear(x: Tensor, sd: SD, name: str) -> Tensor:
    if _FP8["on"] and _is_block_linear(name) and x.shape[-1] % 256 == 0:
        xq, sx = quantize_rows_e4m3(x)
        wq, sw = quantize_rows_e4m3(sd[name + ".weight"])
        y = ((xq @ wq.T) * sx) * sw.squeeze(-1)     # row scale first, then channel scale (the engine's order)
        b = sd.get(name + ".bias")
        return (y + b.float() if b is not None else y).to(x.dtype)
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


# --------------------------------------------------------------------------- embeddings
def timestep_embedding(t: Tensor, dim: int = 256) -> Tensor:
    """get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0,
    scale=1, max_period=10000 (D/models/embeddings.py:27-78 as configured by
    CombinedTimestepGuidanceTextProjEmbeddings, :1322).  fp32 result [N, dim]
    laid out (cos | sin)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    ang = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def time_text_embed(sd: SD, timestep: Tensor, guidance: Optional[Tensor], pooled: Tensor) -> Tensor:
    """CombinedTimestepGuidanceTextProjEmbeddings.forward (D/models/embeddings.py:1327-1339);
    TimestepEmbedding = linear_1 -> SiLU -> linear_2 (:1008-1021); PixArtAlphaTextProjection
    with act silu (:1926-1931)."""
    p = "time_text_embed."
    te = timestep_embedding(timestep).to(pooled.dtype)
    emb = linear(F.silu(linear(te, sd, p + "timestep_embedder.linear_1")), sd, p + "timestep_embedder.linear_2")
    if guidance is not None:
        ge = timestep_embedding(guidance).to(pooled.dtype)
        g = linear(F.silu(linear(ge, sd, p + "guidance_embedder.linear_1")), sd, p + "guidance_embedder.linear_2")
        emb = emb + g
    pe = linear(F.silu(linear(pooled, sd, p + "text_embedder.linear_1")), sd, p + "text_embedder.linear_2")
    return emb + pe


def rope_1d(dim: int, pos: Tensor, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """get_1d_rotary_pos_embed(use_real=True, repeat_interleave_real=True, float64 freqs)
    (D/models/embeddings.py:813-876)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
    ang = torch.outer(pos.to(torch.float64), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def flux_pos_embed(ids: Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """FluxPosEmbed.forward (D/models/embeddings.py:953-973): ids [N,3] -> (cos, sin) fp32 [N, sum(axes)]."""
    pos = ids.float()
    cs = [rope_1d(axes_dim[i], pos[:, i], theta) for i in range(ids.shape[-1])]
    return torch.cat([c for c, _ in cs], dim=-1), torch.cat([s for _, s in cs], dim=-1)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """apply_rotary_emb(use_real=True, use_real_unbind_dim=-1) (D/models/embeddings.py:899-918).
    x [B,H,N,d]; pairs (x0,x1) -> (-x1,x0); fp32 math, one rounding back to x.dtype."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


# --------------------------------------------------------------------------- norms
def rms_norm(x: Tensor, weight: Tensor, eps: float = 1e-6) -> Tensor:
    """RMSNorm.forward (D/models/normalization.py:534-549): fp32 variance, x*rsqrt promoted,
    cast to the weight dtype when that is half precision, then * weight."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        h = h.to(weight.dtype)
    return h * weight


def layer_norm(x: Tensor, eps: float = 1e-6) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def ada_ln_zero(sd: SD, prefix: str, x: Tensor, temb: Tensor):
    """AdaLayerNormZero.forward (D/models/normalization.py:158-171): chunk order
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp."""
    emb = linear(F.silu(temb), sd, prefix + ".linear")
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
    x = layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


def ada_ln_zero_single(sd: SD, prefix: str, x: Tensor, temb: Tensor):
    """AdaLayerNormZeroSingle.forward (D/models/normalization.py:195-203): shift, scale, gate."""
    emb = linear(F.silu(temb), sd, prefix + ".linear")
    shift, scale, gate = emb.chunk(3, dim=1)
    return layer_norm(x) * (1 + scale[:, None]) + shift[:, None], gate


def ada_ln_continuous(sd: SD, prefix: str, x: Tensor, temb: Tensor) -> Tensor:
    """AdaLayerNormContinuous.forward (D/models/normalization.py:361-366): chunk order scale, shift."""
    emb = linear(F.silu(temb).to(x.dtype), sd, prefix + ".linear")
    scale, shift = torch.chunk(emb, 2, dim=1)
    return layer_norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


# --------------------------------------------------------------------------- attention / MLP
def feed_forward(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """FeedForward(activation_fn='gelu-approximate') (D/models/attention.py:1217-1243):
    net.0.proj -> GELU(tanh) (D/models/activations.py:83) -> net.2."""
    return linear(F.gelu(linear(x, sd, prefix + ".net.0.proj"), approximate="tanh"), sd, prefix + ".net.2")


def _heads(x: Tensor, H: int) -> Tensor:
    B, n, D = x.shape
    return x.view(B, n, H, D // H).transpose(1, 2)


def flux_attention(sd: SD, prefix: str, H: int, hidden: Tensor, enc: Optional[Tensor], cos: Tensor, sin: Tensor):
    """FluxAttnProcessor2_0.__call__ (D/models/attention_processor.py:1979-2060)."""
    q = rms_norm(_heads(linear(hidden, sd, prefix + ".to_q"), H), sd[prefix + ".norm_q.weight"])
    k = rms_norm(_heads(linear(hidden, sd, prefix + ".to_k"), H), sd[prefix + ".norm_k.weight"])
    v = _heads(linear(hidden, sd, prefix + ".to_v"), H)
    if enc is not None:
        eq = rms_norm(_heads(linear(enc, sd, prefix + ".add_q_proj"), H), sd[prefix + ".norm_added_q.weight"])
        ek = rms_norm(_heads(linear(enc, sd, prefix + ".add_k_proj"), H), sd[prefix + ".norm_added_k.weight"])
        ev = _heads(linear(enc, sd, prefix + ".add_v_proj"), H)
        q, k, v = torch.cat([eq, q], dim=2), torch.cat([ek, k], dim=2), torch.cat([ev, v], dim=2)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    B = o.shape[0]
    o = o.transpose(1, 2).reshape(B, -1, q.shape[1] * q.shape[-1]).to(q.dtype)
    if enc is not None:
        T = enc.shape[1]
        eo, o = o[:, :T], o[:, T:]
        return linear(o, sd, prefix + ".to_out.0"), linear(eo, sd, prefix + ".to_add_out")
    return o


def double_block(sd: SD, prefix: str, H: int, hidden: Tensor, enc: Tensor, temb: Tensor, cos: Tensor, sin: Tensor):
    """FluxTransformerBlock.forward (D/models/transformers/transformer_flux.py:794-841)."""
    nh, gate_msa, shift_mlp, scale_mlp, gate_mlp = ada_ln_zero(sd, prefix + ".norm1", hidden, temb)
    ne, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = ada_ln_zero(sd, prefix + ".norm1_context", enc, temb)
    attn_out, ctx_out = flux_attention(sd, prefix + ".attn", H, nh, ne, cos, sin)
    hidden = hidden + gate_msa.unsqueeze(1) * attn_out
    nh = layer_norm(hidden) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    hidden = hidden + gate_mlp.unsqueeze(1) * feed_forward(sd, prefix + ".ff", nh)
    enc = enc + c_gate_msa.unsqueeze(1) * ctx_out
    ne = layer_norm(enc) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
    enc = enc + c_gate_mlp.unsqueeze(1) * feed_forward(sd, prefix + ".ff_context", ne)
    return enc, hidden


def single_block(sd: SD, prefix: str, H: int, hidden: Tensor, temb: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """FluxSingleTransformerBlock.forward (D/models/transformers/transformer_flux.py:715-739)."""
    residual = hidden
    nh, gate = ada_ln_zero_single(sd, prefix + ".norm", hidden, temb)
    mlp = F.gelu(linear(nh, sd, prefix + ".proj_mlp"), approximate="tanh")
    attn = flux_attention(sd, prefix + ".attn", H, nh, None, cos, sin)
    out = gate.unsqueeze(1) * linear(torch.cat([attn, mlp], dim=2), sd, prefix + ".proj_out")
    return residual + out


def transformer_forward(
    sd: SD,
    cfg: FluxConfig,
    hidden_states: Tensor,          # [B,S,in_channels]
    encoder_hidden_states: Tensor,  # [B,T,joint_dim]
    pooled_projections: Tensor,     # [B,pooled_dim]
    timestep: Tensor,               # [B] (already /1000, in model dtype as the pipeline passes it)
    img_ids: Tensor,                # [S,3]
    txt_ids: Tensor,                # [T,3]
    guidance: Optional[Tensor] = None,
) -> Tensor:
    """FluxTransformer2DModel.forward (D/models/transformers/transformer_flux.py:1028-1212)."""
    H = cfg.num_attention_heads
    hidden = linear(hidden_states, sd, "x_embedder")
    timestep = timestep.to(hidden.dtype) * 1000
    if guidance is not None:
        guidance = guidance.to(hidden.dtype) * 1000
    temb = time_text_embed(sd, timestep, guidance if cfg.guidance_embeds else None, pooled_projections)
    enc = linear(encoder_hidden_states, sd, "context_embedder")
    cos, sin = flux_pos_embed(torch.cat((txt_ids, img_ids), dim=0), cfg.axes_dims_rope)
    for i in range(cfg.num_layers):
        enc, hidden = double_block(sd, f"transformer_blocks.{i}", H, hidden, enc, temb, cos, sin)
    hidden = torch.cat([enc, hidden], dim=1)
    for i in range(cfg.num_single_layers):
        hidden = single_block(sd, f"single_transformer_blocks.{i}", H, hidden, temb, cos, sin)
    hidden = hidden[:, enc.shape[1]:]
    hidden = ada_ln_continuous(sd, "norm_out", hidden, temb)
    return linear(hidden, sd, "proj_out")


# --------------------------------------------------------------------------- seeded weights
def state_dict_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape table of FluxTransformer2DModel.state_dict(),
    in the order the reference module registers its parameters."""
    D, J, P = cfg.inner_dim, cfg.joint_attention_dim, cfg.pooled_projection_dim
    d = cfg.attention_head_dim
    out: Dict[str, Tuple[int, ...]] = {}

    def lin(name, o, i):
        out[name + ".weight"] = (o, i)
        out[name + ".bias"] = (o,)

    for e in ("timestep_embedder",) + (("guidance_embedder",) if cfg.guidance_embeds else ()):
        lin(f"time_text_embed.{e}.linear_1", D, 256)
        lin(f"time_text_embed.{e}.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, P)
    lin("time_text_embed.text_embedder.linear_2", D, D)
    lin("context_embedder", D, J)
    lin("x_embedder", D, cfg.in_channels)
    for n in range(cfg.num_layers):
        p = f"transformer_blocks.{n}"
        lin(p + ".norm1.linear", 6 * D, D)
        lin(p + ".norm1_context.linear", 6 * D, D)
        out[p + ".attn.norm_q.weight"] = (d,)
        out[p + ".attn.norm_k.weight"] = (d,)
        for w in ("to_q", "to_k", "to_v", "add_k_proj", "add_v_proj", "add_q_proj", "to_out.0", "to_add_out"):
            lin(p + ".attn." + w, D, D)
        out[p + ".attn.norm_added_q.weight"] = (d,)
        out[p + ".attn.norm_added_k.weight"] = (d,)
        lin(p + ".ff.net.0.proj", 4 * D, D)
        lin(p + ".ff.net.2", D, 4 * D)
        lin(p + ".ff_context.net.0.proj", 4 * D, D)
        lin(p + ".ff_context.net.2", D, 4 * D)
    for n in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{n}"
        lin(p + ".norm.linear", 3 * D, D)
        lin(p + ".proj_mlp", 4 * D, D)
        lin(p + ".proj_out", D, 5 * D)
        out[p + ".attn.norm_q.weight"] = (d,)
        out[p + ".attn.norm_k.weight"] = (d,)
        for w in ("to_q", "to_k", "to_v"):
            lin(p + ".attn." + w, D, D)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg.patch_size * cfg.patch_size * cfg.out_channels, D)
    return out


def seeded_state_dict(cfg: FluxConfig, seed: int = 0, dtype=torch.float32, w_std: float = 0.02,
                      b_std: float = 0.02, mod_std: float = 0.02) -> SD:
    """Deterministic synthetic weights (no real checkpoints in this environment).
    One CPU generator, keys visited in state_dict order: weights ~ N(0, w_std), biases ~ N(0, b_std),
    RMSNorm scales ~ 1 + N(0, 0.1).  Non-zero biases / non-unit norm scales so that a kernel that
    drops them cannot pass parity."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for k, shape in state_dict_shapes(cfg).items():
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if ".norm_" in k and k.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * r
        elif k.endswith(".bias"):
            t = b_std * r
        elif ".linear.weight" in k and "norm" in k:
            t = mod_std * r
        else:
            t = w_std * r
        sd[k] = t.to(dtype)
    return sd
