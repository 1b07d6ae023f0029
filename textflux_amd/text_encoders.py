"""The two text encoders of FluxFillPipeline.encode_prompt on the gfx950 kernels.

Reference call sites: `self.text_encoder_2(ids, output_hidden_states=False)[0]` (T5-XXL sequence embedding,
D/pipelines/flux/pipeline_flux_fill.py:1447) and `self.text_encoder(ids, output_hidden_states=False).pooler_output` (CLIP-L
pooled embedding, :1493-1496).  Their arithmetic is third-party `transformers` (pinned 4.43.3, not under /root/reference:
models/t5/modeling_t5.py, models/clip/modeling_clip.py); the classes below keep those two call signatures and read the same
checkpoint layout (config.json + model.safetensors, or the sharded form with model.safetensors.index.json), and run the
arithmetic in libtextflux_hip.so:
  Linear layers      tfx_gemm_bf16 (q | k | v and wi_1 | wi_0 fused by row concatenation at load, GELU-tanh in the epilogue,
                     the residual add of `o` / `wo` / out_proj / fc2 in the epilogue: Linear output rounded to bf16, then the
                     bf16 sum -- the reference's rounding points)
  residual stream    bf16 in both models.  T5's `_keep_in_fp32_modules = ["wo"]` only applies to a float16 load (transformers
                     modeling_utils, 4.43.3 and 5.x alike); the reference loads with torch_dtype=bfloat16, so `wo` is bf16 and
                     the stream is rounded to bf16 after every sublayer
  attention          tfx_attention64 (heads of 64; T5: unscaled scores + bucketed relative-position bias; CLIP: causal)
  norms              tfx_rmsnorm (T5LayerNorm), tfx_layernorm (nn.LayerNorm with its affine, one rounding, any gamma)
  embeddings etc.    tfx_gather_rows, tfx_add, tfx_mul_act
Tokenizers stay `transformers` objects (host string processing).  Parity: oracle/text_oracle.py restates both models and is
pinned against `transformers` (tests/test_text_oracle.py); tests/test_text_encoders_gpu.py compares these classes with it.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import ops

BF16 = torch.bfloat16


def _load_safetensors_dir(root: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    index = os.path.join(root, "model.safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = ["model.safetensors"]
    sd = {}
    for fn in files:
        with safe_open(os.path.join(root, fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    return sd


class _Encoder:
    dtype = BF16

    def __init__(self):
        self.device = torch.device("cpu")
        self.w: Dict[str, torch.Tensor] = {}

    def to(self, device=None, dtype=None):
        if dtype is not None and dtype != BF16:
            raise ValueError("the HIP text encoders compute in bf16")
        if device is not None and torch.device(device) != self.device:
            self.w = {k: v.to(device) for k, v in self.w.items()}
            self.device = torch.device(device)
        return self

    def eval(self):
        return self

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype=BF16, device="cuda", **_):
        if torch_dtype not in (BF16, None):
            raise ValueError("the HIP text encoders compute in bf16; pass torch_dtype=torch.bfloat16")
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "config.json")) as f:
            cfg = json.load(f)
        return cls(cfg).load_state_dict(_load_safetensors_dir(root), device=device)


class T5EncoderModel(_Encoder):
    """T5 v1.1 encoder stack (gated-GELU feed forward, relative-position bias in layer 0 shared by all layers)."""

    def __init__(self, config: Dict):
        super().__init__()
        c = config if isinstance(config, dict) else config.to_dict()
        if c.get("feed_forward_proj", "gated-gelu") != "gated-gelu" or c.get("d_kv", 64) != 64:
            raise ValueError("T5EncoderModel on the gfx950 kernels: gated-gelu feed forward and d_kv = 64 (T5 v1.1 / FLUX's T5-XXL)")
        self.config = SimpleNamespace(d_model=c["d_model"], d_kv=64, num_heads=c["num_heads"], d_ff=c["d_ff"],
                                      num_layers=c["num_layers"], vocab_size=c["vocab_size"],
                                      relative_attention_num_buckets=c.get("relative_attention_num_buckets", 32),
                                      relative_attention_max_distance=c.get("relative_attention_max_distance", 128),
                                      layer_norm_epsilon=c.get("layer_norm_epsilon", 1e-6))
        self._bias_tables: Dict[int, torch.Tensor] = {}

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device="cuda"):
        c = self.config
        dev = torch.device(device)
        g = lambda k: sd[k].to(dev, BF16)
        emb = "shared.weight" if "shared.weight" in sd else "encoder.embed_tokens.weight"
        w = {"embed": g(emb).contiguous(), "final_ln": g("encoder.final_layer_norm.weight"),
             "rel_bias": sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].to(dev, BF16)}
        for i in range(c.num_layers):
            p, a, f = f"encoder.block.{i}.layer.", f"encoder.block.{i}.layer.0.SelfAttention.", f"encoder.block.{i}.layer.1.DenseReluDense."
            w[f"{i}.ln0"], w[f"{i}.ln1"] = g(p + "0.layer_norm.weight"), g(p + "1.layer_norm.weight")
            w[f"{i}.qkv"] = torch.cat([g(a + "q.weight"), g(a + "k.weight"), g(a + "v.weight")], 0).contiguous()
            w[f"{i}.o"] = g(a + "o.weight").contiguous()
            w[f"{i}.wi"] = torch.cat([g(f + "wi_1.weight"), g(f + "wi_0.weight")], 0).contiguous()   # [linear ; gelu] rows
            w[f"{i}.wo"] = g(f + "wo.weight").contiguous()
        self.w, self.device = w, dev
        self._bias_tables = {}
        return self

    def init_random_(self, seed: int = 0, device="cuda", std: float = 0.02):
        """Random weights of the configured shapes, generated on the device (synthetic benchmarking; no checkpoints offline)."""
        c = self.config
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        inner = c.num_heads * 64

        def rnd(*shape, scale=std):
            t = torch.empty(*shape, dtype=BF16, device=dev)
            flat = t.view(-1)
            for s0 in range(0, flat.numel(), 1 << 26):
                e = min(flat.numel(), s0 + (1 << 26))
                flat[s0:e].copy_((torch.randn(e - s0, generator=g, device=dev) * scale).to(BF16))
            return t

        ones = lambda n: (1.0 + 0.1 * torch.randn(n, generator=g, device=dev)).to(BF16)
        w = {"embed": rnd(c.vocab_size, c.d_model, scale=1.0), "final_ln": ones(c.d_model),
             "rel_bias": rnd(c.relative_attention_num_buckets, c.num_heads, scale=0.5)}
        for i in range(c.num_layers):
            w[f"{i}.ln0"], w[f"{i}.ln1"] = ones(c.d_model), ones(c.d_model)
            w[f"{i}.qkv"], w[f"{i}.o"] = rnd(3 * inner, c.d_model), rnd(c.d_model, inner)
            w[f"{i}.wi"], w[f"{i}.wo"] = rnd(2 * c.d_ff, c.d_model), rnd(c.d_model, c.d_ff)
        self.w, self.device, self._bias_tables = w, dev, {}
        return self

    def _rel_bias(self, T: int) -> torch.Tensor:
        """fp32 [H, 2T - 1]: bias(h, key - query) (T5Attention.compute_bias depends on the distance only).  Host integer
        bucketing of the 2T - 1 distances, one table lookup."""
        t = self._bias_tables.get(T)
        if t is None:
            import math
            c = self.config
            rel = torch.arange(-(T - 1), T)
            nb = c.relative_attention_num_buckets // 2
            ret = (rel > 0).long() * nb
            rp = rel.abs()
            max_exact = nb // 2
            large = max_exact + (torch.log(rp.float() / max_exact) / math.log(c.relative_attention_max_distance / max_exact)
                                 * (nb - max_exact)).long()
            large = torch.min(large, torch.full_like(large, nb - 1))
            bucket = ret + torch.where(rp < max_exact, rp, large)
            t = self.w["rel_bias"][bucket.to(self.device)].float().t().contiguous()
            self._bias_tables[T] = t
        return t

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, attention_mask=None, output_hidden_states: bool = False, **_):
        if attention_mask is not None:
            raise NotImplementedError("the Fill pipeline encodes without an attention mask (P:1447)")
        c, w = self.config, self.w
        if input_ids.shape[1] > 512:
            raise ValueError("at most 512 tokens (max_sequence_length of the pipeline)")
        ids = input_ids.to(self.device, torch.int64)
        B, T = ids.shape
        D, inner, dff = c.d_model, c.num_heads * 64, c.d_ff
        x = ops.gather_rows(w["embed"], ids)                                   # [B * T, D] bf16: the residual stream
        bias = self._rel_bias(T)
        for i in range(c.num_layers):
            h = ops.rmsnorm(x, w[f"{i}.ln0"], c.layer_norm_epsilon)
            qkv = ops.gemm(h, w[f"{i}.qkv"], None).view(B, T, 3 * inner)
            a = ops.attention64(qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:], 1.0, rel_bias=bias)
            x = ops.gemm(a.view(B * T, inner), w[f"{i}.o"], None, epilogue=ops.EPI_BIAS_RES, res=x)
            h = ops.rmsnorm(x, w[f"{i}.ln1"], c.layer_norm_epsilon)
            u = ops.gemm(h, w[f"{i}.wi"], None, epilogue=ops.EPI_BIAS_GELU, gelu_from_col=dff)       # [wi_1 x | gelu(wi_0 x)]
            x = ops.gemm(ops.mul(u[:, dff:], u[:, :dff]), w[f"{i}.wo"], None, epilogue=ops.EPI_BIAS_RES, res=x)
        out = ops.rmsnorm(x, w["final_ln"], c.layer_norm_epsilon).view(B, T, D)
        return (out,)


class CLIPTextModel(_Encoder):
    """CLIP text transformer (pre-LN, causal attention, quick_gelu MLP), pooled at the EOS token."""

    def __init__(self, config: Dict):
        super().__init__()
        c = config if isinstance(config, dict) else config.to_dict()
        if "text_config" in c and "hidden_size" not in c:
            c = c["text_config"]
        if c.get("hidden_act", "quick_gelu") != "quick_gelu" or c["hidden_size"] // c["num_attention_heads"] != 64:
            raise ValueError("CLIPTextModel on the gfx950 kernels: quick_gelu MLP and heads of dim 64 (CLIP ViT-L/14 text model)")
        self.config = SimpleNamespace(hidden_size=c["hidden_size"], num_attention_heads=c["num_attention_heads"],
                                      intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
                                      max_position_embeddings=c.get("max_position_embeddings", 77), vocab_size=c["vocab_size"],
                                      layer_norm_eps=c.get("layer_norm_eps", 1e-5), eos_token_id=c.get("eos_token_id", 2))

    def load_state_dict(self, sd: Dict[str, torch.Tensor], device="cuda"):
        c = self.config
        dev = torch.device(device)
        p = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
        g = lambda k: sd[p + k].to(dev, BF16).contiguous()

        def ln(name, key):   # nn.LayerNorm's own (gamma, beta): any value loads (tfx_layernorm)
            w[name + ".gamma"], w[name + ".beta"] = g(key + ".weight"), g(key + ".bias")

        w: Dict[str, torch.Tensor] = {"tok": g("embeddings.token_embedding.weight"), "pos": g("embeddings.position_embedding.weight")}
        for i in range(c.num_hidden_layers):
            l = f"encoder.layers.{i}."
            ln(f"{i}.ln1", l + "layer_norm1")
            ln(f"{i}.ln2", l + "layer_norm2")
            w[f"{i}.qkv.w"] = torch.cat([g(l + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous()
            w[f"{i}.qkv.b"] = torch.cat([g(l + f"self_attn.{n}_proj.bias") for n in "qkv"], 0).contiguous()
            for n, k in (("o", "self_attn.out_proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                w[f"{i}.{n}.w"], w[f"{i}.{n}.b"] = g(l + k + ".weight"), g(l + k + ".bias")
        ln("final", "final_layer_norm")
        self.w, self.device = w, dev
        return self

    def init_random_(self, seed: int = 0, device="cuda", std: float = 0.02):
        c = self.config
        dev = torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        rnd = lambda *shape, scale=std: (torch.randn(*shape, generator=g, device=dev) * scale).to(BF16)
        D, I = c.hidden_size, c.intermediate_size
        w = {"tok": rnd(c.vocab_size, D, scale=1.0), "pos": rnd(c.max_position_embeddings, D, scale=0.1)}
        names = [f"{i}.{n}" for i in range(c.num_hidden_layers) for n in ("ln1", "ln2")] + ["final"]
        for n in names:
            w[n + ".gamma"], w[n + ".beta"] = (1.0 + rnd(D, scale=0.05).float()).to(BF16), rnd(D, scale=0.05)
        for i in range(c.num_hidden_layers):
            w[f"{i}.qkv.w"], w[f"{i}.qkv.b"] = rnd(3 * D, D), rnd(3 * D)
            w[f"{i}.o.w"], w[f"{i}.o.b"] = rnd(D, D), rnd(D)
            w[f"{i}.fc1.w"], w[f"{i}.fc1.b"] = rnd(I, D), rnd(I)
            w[f"{i}.fc2.w"], w[f"{i}.fc2.b"] = rnd(D, I), rnd(D)
        self.w, self.device = w, dev
        return self

    def _ln(self, x, name):
        return ops.layernorm(x.contiguous(), self.w[name + ".gamma"], self.w[name + ".beta"], eps=self.config.layer_norm_eps)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, attention_mask=None, output_hidden_states: bool = False, **_):
        if attention_mask is not None:
            raise NotImplementedError("the Fill pipeline encodes without an attention mask (P:1493)")
        c, w = self.config, self.w
        ids = input_ids.to(self.device, torch.int64)
        B, T = ids.shape
        if T > c.max_position_embeddings:
            raise ValueError(f"sequence length {T} exceeds max_position_embeddings {c.max_position_embeddings}")
        D = c.hidden_size
        pos = torch.arange(T, device=self.device, dtype=torch.int64).repeat(B)
        x = ops.add(ops.gather_rows(w["tok"], ids), ops.gather_rows(w["pos"], pos)).view(B, T, D)
        for i in range(c.num_hidden_layers):
            h = self._ln(x, f"{i}.ln1")
            qkv = ops.gemm(h, w[f"{i}.qkv.w"], w[f"{i}.qkv.b"])
            a = ops.attention64(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], 64 ** -0.5, causal=True)
            x = ops.gemm(a, w[f"{i}.o.w"], w[f"{i}.o.b"], epilogue=ops.EPI_BIAS_RES, res=x)
            h = self._ln(x, f"{i}.ln2")
            u = ops.quick_gelu(ops.gemm(h, w[f"{i}.fc1.w"], w[f"{i}.fc1.b"]).view(B * T, -1)).view(B, T, -1)
            x = ops.gemm(u, w[f"{i}.fc2.w"], w[f"{i}.fc2.b"], epilogue=ops.EPI_BIAS_RES, res=x)
        last = self._ln(x, "final")
        eos = ids.argmax(dim=-1) if c.eos_token_id == 2 else (ids == c.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=self.device), eos]
        return SimpleNamespace(last_hidden_state=last, pooler_output=pooled)
