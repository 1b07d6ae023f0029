"""FluxTransformer2DModel on the MI355X HIP engine.

Drop-in for the reference class at the `forward(...)` / `from_pretrained(...)` / `.config` level
(reference: diffusers/src/diffusers/models/transformers/transformer_flux.py:848-1212, "D/" below), plus an
engine-level session API the pipeline uses to hoist everything that does not depend on the denoising step:

* context_embedder(prompt_embeds)                      (D/.../transformer_flux.py:1099)  once per call
* RoPE tables from txt/img ids                          (:1114-1115)                      once per call
* time/guidance/pooled embedding -> temb -> SiLU -> every AdaLN modulation Linear of every block, as ONE
  GEMM against the row-stacked modulation matrix, for ALL steps at once (:1094-1098 and
  D/models/normalization.py:168, 200, 364)

Weights are stored pre-fused (pure row concatenations): [to_k; to_v; to_q] per stream,
[to_k; to_v; to_q; proj_mlp] per single block, and all norm*.linear stacked.  All arithmetic runs in
libtextflux_hip.so; torch only owns the device buffers.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops

BF16 = torch.bfloat16
EULER_PAD = 64     # columns appended to a step's modulation rows when the Euler update rides on proj_out's epilogue (= out_channels)


def rope_tables(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [N, sum(axes_dim)] fp32, float64 frequencies, each value repeated twice (interleaved pairs)
    (FluxPosEmbed.forward, D/models/embeddings.py:953-973; get_1d_rotary_pos_embed :853-866).  Host logic."""
    pos = ids.detach().to("cpu", torch.float64)
    cos, sin = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i], freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=-1).contiguous(), torch.cat(sin, dim=-1).contiguous()


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)


class FluxTransformer2DModel:
    config_name = "config.json"

    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: Optional[int] = None,
                 num_layers: int = 19, num_single_layers: int = 38, attention_head_dim: int = 128,
                 num_attention_heads: int = 24, joint_attention_dim: int = 4096, pooled_projection_dim: int = 768,
                 guidance_embeds: bool = False, axes_dims_rope: Tuple[int, ...] = (16, 56, 56)):
        if attention_head_dim != 128:
            raise ValueError("the gfx950 attention / RMSNorm+RoPE kernels are specialised for head_dim 128 (FLUX.1)")
        if sum(axes_dims_rope) != attention_head_dim:
            raise ValueError("sum(axes_dims_rope) must equal attention_head_dim")
        if patch_size != 1:
            raise ValueError("patch_size must be 1 (FLUX.1)")
        self.config = _Config(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
                              num_layers=num_layers, num_single_layers=num_single_layers,
                              attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                              joint_attention_dim=joint_attention_dim, pooled_projection_dim=pooled_projection_dim,
                              guidance_embeds=guidance_embeds, axes_dims_rope=tuple(axes_dims_rope))
        self.out_channels = out_channels or in_channels
        self.inner_dim = num_attention_heads * attention_head_dim
        self.dtype = BF16
        self.device = torch.device("cpu")
        self.w: Dict[str, torch.Tensor] = {}     # fused weights (see module docstring)
        self.w8: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}   # enable_fp8(): name -> (e4m3 bytes [out,in], scale f32 [out])
        self._session: Optional["DitSession"] = None
        self.fuse_qk_norm_rope = True   # q/k RMSNorm + RoPE in the projection GEMM's epilogue where eligible (tfx_dit_desc.rope_cs)
        D = self.inner_dim
        self.mod_len = 12 * D * num_layers + 3 * D * num_single_layers + 2 * D

    # ------------------------------------------------------------------ weights
    def _alloc(self, device) -> None:
        c, D = self.config, self.inner_dim

        def lin(name, o, i):
            self.w[name + ".w"] = torch.empty(o, i, dtype=BF16, device=device)
            self.w[name + ".b"] = torch.empty(o, dtype=BF16, device=device)

        embs = ["timestep_embedder"] + (["guidance_embedder"] if c.guidance_embeds else [])
        for e in embs:
            lin(f"{e}.1", D, 256)
            lin(f"{e}.2", D, D)
        lin("text_embedder.1", D, c.pooled_projection_dim)
        lin("text_embedder.2", D, D)
        lin("context_embedder", D, c.joint_attention_dim)
        lin("x_embedder", D, c.in_channels)
        lin("proj_out", self.out_channels, D)
        lin("mod", self.mod_len, D)
        def lin_pair(img, txt, o, i):
            # the image- and text-stream Linear of a double block side by side in ONE allocation ([img; txt]): a single buffer
            # descriptor reaches both, which is what lets the engine run the two projections as one launch over the joint
            # [text | image] rows (row-split weights, tfx_dit_forward); every other user sees two ordinary tensors
            wp, bp = torch.empty(2, o, i, dtype=BF16, device=device), torch.empty(2, o, dtype=BF16, device=device)
            self.w[img + ".w"], self.w[img + ".b"] = wp[0], bp[0]
            self.w[txt + ".w"], self.w[txt + ".b"] = wp[1], bp[1]

        for i in range(c.num_layers):
            lin_pair(f"d{i}.qkv_img", f"d{i}.qkv_txt", 3 * D, D)
            lin_pair(f"d{i}.out_img", f"d{i}.out_txt", D, D)
            lin_pair(f"d{i}.ff1_img", f"d{i}.ff1_txt", 4 * D, D)
            lin_pair(f"d{i}.ff2_img", f"d{i}.ff2_txt", D, 4 * D)
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                self.w[f"d{i}.{n}"] = torch.empty(128, dtype=BF16, device=device)
        for j in range(c.num_single_layers):
            lin(f"s{j}.qkv_mlp", 7 * D, D)
            lin(f"s{j}.proj_out", D, 5 * D)
            for n in ("norm_q", "norm_k"):
                self.w[f"s{j}.{n}"] = torch.empty(128, dtype=BF16, device=device)
        self.device = torch.device(device)

    def _fusion_map(self) -> List[Tuple[str, str, int]]:
        """(reference key prefix, fused tensor name, row offset) for every nn.Linear of the reference state dict."""
        c, D = self.config, self.inner_dim
        m: List[Tuple[str, str, int]] = []
        embs = ["timestep_embedder"] + (["guidance_embedder"] if c.guidance_embeds else [])
        for e in embs:
            m += [(f"time_text_embed.{e}.linear_1", f"{e}.1", 0), (f"time_text_embed.{e}.linear_2", f"{e}.2", 0)]
        m += [("time_text_embed.text_embedder.linear_1", "text_embedder.1", 0),
              ("time_text_embed.text_embedder.linear_2", "text_embedder.2", 0),
              ("context_embedder", "context_embedder", 0), ("x_embedder", "x_embedder", 0), ("proj_out", "proj_out", 0)]
        for i in range(c.num_layers):
            p = f"transformer_blocks.{i}"
            m += [(p + ".norm1.linear", "mod", 12 * D * i), (p + ".norm1_context.linear", "mod", 12 * D * i + 6 * D)]
            m += [(p + ".attn.to_k", f"d{i}.qkv_img", 0), (p + ".attn.to_v", f"d{i}.qkv_img", D),
                  (p + ".attn.to_q", f"d{i}.qkv_img", 2 * D)]
            m += [(p + ".attn.add_k_proj", f"d{i}.qkv_txt", 0), (p + ".attn.add_v_proj", f"d{i}.qkv_txt", D),
                  (p + ".attn.add_q_proj", f"d{i}.qkv_txt", 2 * D)]
            m += [(p + ".attn.to_out.0", f"d{i}.out_img", 0), (p + ".attn.to_add_out", f"d{i}.out_txt", 0),
                  (p + ".ff.net.0.proj", f"d{i}.ff1_img", 0), (p + ".ff.net.2", f"d{i}.ff2_img", 0),
                  (p + ".ff_context.net.0.proj", f"d{i}.ff1_txt", 0), (p + ".ff_context.net.2", f"d{i}.ff2_txt", 0)]
        base = 12 * D * c.num_layers
        for j in range(c.num_single_layers):
            p = f"single_transformer_blocks.{j}"
            m += [(p + ".norm.linear", "mod", base + 3 * D * j)]
            m += [(p + ".attn.to_k", f"s{j}.qkv_mlp", 0), (p + ".attn.to_v", f"s{j}.qkv_mlp", D),
                  (p + ".attn.to_q", f"s{j}.qkv_mlp", 2 * D), (p + ".proj_mlp", f"s{j}.qkv_mlp", 3 * D),
                  (p + ".proj_out", f"s{j}.proj_out", 0)]
        m += [("norm_out.linear", "mod", base + 3 * D * c.num_single_layers)]
        return m

    def _norm_map(self) -> List[Tuple[str, str]]:
        c = self.config
        m = []
        for i in range(c.num_layers):
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                m.append((f"transformer_blocks.{i}.attn.{n}.weight", f"d{i}.{n}"))
        for j in range(c.num_single_layers):
            for n in ("norm_q", "norm_k"):
                m.append((f"single_transformer_blocks.{j}.attn.{n}.weight", f"s{j}.{n}"))
        return m

    def expected_keys(self) -> List[str]:
        return [k + s for k, _, _ in self._fusion_map() for s in (".weight", ".bias")] + [k for k, _ in self._norm_map()]

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, device=None):
        """Load a reference-format state dict (SURVEY.md Appendix A key names) into the fused device layout."""
        device = device or (self.device if self.device.type != "cpu" else "cuda")
        missing = [k for k in self.expected_keys() if k not in sd]
        unexpected = [k for k in sd if k not in set(self.expected_keys())]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        if not self.w:
            self._alloc(device)
        for key, name, off in self._fusion_map():
            # weight and bias are looked up independently: HF shards split on cumulative size, so a Linear's two
            # tensors may arrive in different shards (from_pretrained calls this once per shard, strict=False)
            wt, bs = sd.get(key + ".weight"), sd.get(key + ".bias")
            if wt is not None:
                self.w[name + ".w"][off:off + wt.shape[0]].copy_(wt.to(BF16))
            if bs is not None:
                self.w[name + ".b"][off:off + bs.shape[0]].copy_(bs.to(BF16))
        for key, name in self._norm_map():
            if key in sd:
                self.w[name].copy_(sd[key].to(BF16))
        self._session = None
        return self

    def init_random_(self, seed: int = 0, device="cuda", w_std: float = 0.02):
        """Random-init fused weights directly on the device (synthetic benchmarking; no checkpoints offline)."""
        if not self.w:
            self._alloc(device)
        g = torch.Generator(device=self.device).manual_seed(seed)
        for k, t in self.w.items():
            if t.dim() == 1 and t.numel() == 128 and ".norm_" in k:
                t.copy_((1 + 0.1 * torch.randn(t.shape, generator=g, device=t.device)).to(BF16))
            else:
                # chunked so that a 1 GB fused matrix never needs a 2 GB fp32 temporary
                flat = t.view(-1)
                for s in range(0, flat.numel(), 1 << 26):
                    e = min(flat.numel(), s + (1 << 26))
                    flat[s:e].copy_((torch.randn(e - s, generator=g, device=t.device) * w_std).to(BF16))
        self._session = None
        return self

    @classmethod
    def from_config(cls, cfg: Dict[str, Any]) -> "FluxTransformer2DModel":
        keys = ("patch_size", "in_channels", "out_channels", "num_layers", "num_single_layers", "attention_head_dim",
                "num_attention_heads", "joint_attention_dim", "pooled_projection_dim", "guidance_embeds",
                "axes_dims_rope")
        return cls(**{k: cfg[k] for k in keys if k in cfg})

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype=BF16, device="cuda", **_):
        """Reads the HF layout: config.json + diffusion_pytorch_model.safetensors or the sharded form with
        diffusion_pytorch_model.safetensors.index.json (D/models/modeling_utils.py:468), streamed straight
        into the fused device layout (loader.ShardStreamer: mmap, pinned double-buffered staging, asynchronous H2D)."""
        from . import loader
        if torch_dtype not in (BF16, None):
            raise ValueError("the HIP engine computes in bf16; pass torch_dtype=torch.bfloat16")
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            model = cls.from_config(json.load(f))
        files = loader.shard_files(root, "diffusion_pytorch_model")
        model._alloc(device)
        # routing table: reference key -> row slice of the fused device matrix it lands in (the concatenation IS the copy)
        route: Dict[str, torch.Tensor] = {}
        for key, name, off in model._fusion_map():
            wt, bs = model.w[name + ".w"], model.w[name + ".b"]
            rows = model._rows_of(key)
            route[key + ".weight"], route[key + ".bias"] = wt[off:off + rows], bs[off:off + rows]
        for key, name in model._norm_map():
            route[key] = model.w[name]
        seen = set()

        def pick(k, shape, dt):
            dst = route.get(k)
            if dst is None:
                return None
            if tuple(dst.shape) != tuple(shape):
                raise RuntimeError(f"{k}: checkpoint shape {tuple(shape)} != model shape {tuple(dst.shape)}")
            seen.add(k)
            return dst

        st = loader.ShardStreamer(device)
        for fn in files:      # shard by shard, tensor by tensor in file order, through pinned double-buffered async copies
            st.stream_file(os.path.join(root, fn), pick)
        st.finish()
        model.load_stats = dict(bytes=st.bytes_moved, files=len(files))
        missing = [k for k in model.expected_keys() if k not in seen]
        if missing:
            raise RuntimeError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        model._session = None
        return model

    def _rows_of(self, key: str) -> int:
        """Output rows of the reference Linear `key` (its weight is [rows, in])."""
        D = self.inner_dim
        if key.endswith("norm1.linear") or key.endswith("norm1_context.linear"):
            return 6 * D
        if key.endswith(".norm.linear"):
            return 3 * D
        if key == "norm_out.linear":
            return 2 * D
        if key.endswith((".net.0.proj", ".proj_mlp")):
            return 4 * D
        if key == "proj_out":
            return self.out_channels
        return D

    def to(self, device=None, dtype=None):
        if dtype is not None and dtype != BF16:
            raise ValueError("the HIP engine computes in bf16")
        if device is not None and self.w and torch.device(device).type != self.device.type:
            self.w = {k: v.to(device) for k, v in self.w.items()}
            self.w8 = {k: (q.to(device), sc.to(device)) for k, (q, sc) in self.w8.items()}
            self.device = torch.device(device)
            self._session = None
        return self

    def eval(self):
        return self

    def fp8_linear_names(self) -> List[str]:
        """The Linears that run in fp8 under enable_fp8(): every GEMM inside the 57 blocks.  Embedders, the modulation
        GEMM and proj_out stay bf16 (K = 384 / once per image / N = 64)."""
        c = self.config
        names = [f"d{i}.{n}" for i in range(c.num_layers)
                 for n in ("qkv_img", "qkv_txt", "out_img", "out_txt", "ff1_img", "ff2_img", "ff1_txt", "ff2_txt")]
        names += [f"s{j}.{n}" for j in range(c.num_single_layers) for n in ("qkv_mlp", "proj_out")]
        return [n for n in names if self.w[n + ".w"].shape[1] % 256 == 0]

    def enable_fp8(self, on: bool = True):
        """BASELINE config 5 (no reference counterpart: the reference computes in bf16): quantise the block weights to
        OCP e4m3 with one scale per output channel (absmax / 448) on the device; activations are quantised per token
        row at run time (tfx_quantize_rows_fp8) and the GEMMs run on the fp8 MFMA with fp32 accumulation, bf16 outputs.
        Call after the weights (and any LoRA merge) are final; the bf16 weights stay resident."""
        if not self.w:
            raise RuntimeError("weights not loaded")
        self.w8 = {}
        if on:
            for n in self.fp8_linear_names():
                q, sc = ops.quantize_rows_fp8(self.w[n + ".w"])
                self.w8[n] = (q, sc)
        self._session = None
        return self

    # ------------------------------------------------------------------ conditioning (step-invariant work)
    def _lin(self, x, name, **kw):
        return ops.gemm(x, self.w[name + ".w"], self.w[name + ".b"], **kw)

    def temb(self, t_f32: torch.Tensor, g_f32: Optional[torch.Tensor], pooled: torch.Tensor) -> torch.Tensor:
        """CombinedTimestepGuidanceTextProjEmbeddings (D/models/embeddings.py:1327-1339).  t_f32 / g_f32: the values
        the sinusoid sees (already x1000 and bf16-rounded by the caller), one per row; pooled [R, P] bf16."""
        te = self._lin(ops.silu(self._lin(ops.timestep_embedding(t_f32), "timestep_embedder.1")), "timestep_embedder.2")
        if self.config.guidance_embeds:
            if g_f32 is None:
                raise ValueError("guidance is required when guidance_embeds=True")
            ge = self._lin(ops.silu(self._lin(ops.timestep_embedding(g_f32), "guidance_embedder.1")),
                           "guidance_embedder.2")
            te = ops.add(te, ge)
        pe = self._lin(ops.silu(self._lin(pooled.contiguous(), "text_embedder.1")), "text_embedder.2")
        return ops.add(te, pe)

    def modulation(self, temb: torch.Tensor) -> torch.Tensor:
        """Every norm*.linear(SiLU(temb)) of the model in one GEMM -> [R, mod_len]."""
        return self._lin(ops.silu(temb), "mod")

    # ------------------------------------------------------------------ sessions / forward
    # 3 % on top of the bound: the normalised q and k each pass three bf16 roundings (x * rsqrt -> bf16, * weight -> bf16, RoPE ->
    # bf16) and q one more when the kernel pre-scales it: (1 + 2^-8)^7 = 1.028 in the worst case.
    SCORE_BOUND_MARGIN = 1.03

    def attn_score_bounds(self):
        """Per block (doubles, then singles): an upper bound of |q . k| * 128^-0.5 of THAT block's attention launch, from its own
        q / k RMSNorm weights (attention_processor.py:2001-2004, 2023-2037: after RMSNorm |q|^2 = sum_i (x_i / rms)^2 w_i^2 <=
        128 max w^2, RoPE is a rotation) -- tfx_double_block / tfx_single_block.attn_score_bound.  It lets the attention kernel drop
        its running-maximum bookkeeping where the bound shows that exp2 cannot leave the exponent range; a block with large norm
        scales falls back alone.  Joint attention mixes the image and text streams: the larger of the two norms on each side.
        One device reduction + one copy for all 57 blocks.  Cached with the weights' version counters: an in-place edit of a norm
        weight re-derives the bounds (and drops the session that baked the old ones) on the next forward."""
        c, w = self.config, self.w
        names = []
        for i in range(c.num_layers):
            names += [f"d{i}.norm_q", f"d{i}.norm_added_q", f"d{i}.norm_k", f"d{i}.norm_added_k"]
        for j in range(c.num_single_layers):
            names += [f"s{j}.norm_q", f"s{j}.norm_k"]
        key = tuple((w[n].data_ptr(), w[n]._version) for n in names)
        cached = getattr(self, "_bounds_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if not names:
            self._bounds_cache = (key, ([], []))
            return [], []
        am = torch.stack([w[n].float().abs().max() for n in names]).cpu().tolist()
        f = self.SCORE_BOUND_MARGIN * 128.0 * 128 ** -0.5
        nd = c.num_layers
        dbl = [f * max(am[4 * i], am[4 * i + 1]) * max(am[4 * i + 2], am[4 * i + 3]) for i in range(nd)]
        sgl = [f * am[4 * nd + 2 * j] * am[4 * nd + 2 * j + 1] for j in range(c.num_single_layers)]
        if cached is not None:
            self._session = None        # a session bakes the bounds into its descriptor (and its captured graphs)
        self._bounds_cache = (key, (dbl, sgl))
        return dbl, sgl

    def attn_score_bound(self) -> float:
        """The largest per-block bound (tfx_dit_desc.attn_score_bound, the forward-wide fallback field)."""
        dbl, sgl = self.attn_score_bounds()
        return max(dbl + sgl, default=0.0)

    def session(self, B: int, S: int, T: int) -> "DitSession":
        self.attn_score_bounds()        # norm weights edited in place since the last call: new bounds, new session
        s = self._session
        if s is None or (s.B, s.S, s.T) != (B, S, T):
            s = self._session = DitSession(self, B, S, T)
        return s

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_block_samples=None,
                controlnet_single_block_samples=None, return_dict: bool = True, controlnet_blocks_repeat: bool = False):
        """Same contract as the reference forward (transformer_flux.py:1028-1212): returns (sample,) when
        return_dict=False, else an object with `.sample`."""
        if controlnet_block_samples is not None or controlnet_single_block_samples is not None:
            raise NotImplementedError("ControlNet residuals are outside the TextFlux hot path (SURVEY.md §2.2)")
        if not self.w:
            raise RuntimeError("weights not loaded")
        dev = self.device
        hs = hidden_states.to(dev, BF16)
        B, S, _ = hs.shape
        T = encoder_hidden_states.shape[1]
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        ses = self.session(B, S, T)
        ses.set_conditioning(encoder_hidden_states.to(dev, BF16), txt_ids, img_ids)
        # timestep.to(dtype) * 1000 and guidance.to(dtype) * 1000 in the model dtype (transformer_flux.py:1088-1090)
        t = (timestep.to(dev, BF16) * 1000).float().expand(B)
        g = (guidance.to(dev, BF16) * 1000).float().expand(B) if guidance is not None else None
        mod = self.modulation(self.temb(t, g, pooled_projections.to(dev, BF16)))
        ses.xin.copy_(hs)
        out = ses.run(mod).clone()
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)

    __call__ = forward


class DitSession:
    """Device workspace + C descriptor for one (B, S, T) problem size."""

    def __init__(self, model: FluxTransformer2DModel, B: int, S: int, T: int):
        self.model, self.B, self.S, self.T = model, B, S, T
        c, D, dev = model.config, model.inner_dim, model.device
        N = S + T
        self.N = N
        e = lambda *shape: torch.empty(*shape, dtype=BF16, device=dev)
        self.xin = e(B, S, c.in_channels)
        self.ctx0 = e(B, T, D)
        self.out = e(B, S, model.out_channels)
        # the engine's workspace is ONE allocation laid out by the library (tfx_workspace_layout): hid | xn | y | q8 | ...
        self.fp8 = bool(model.w8)
        flags = 4 if self.fp8 else 0
        off, gws = (C.c_int64 * 6)(), C.c_int64()
        L.check(L.lib().tfx_workspace_layout(B, S, T, D, flags, off, C.byref(gws)), "workspace_layout")
        total = L.lib().tfx_workspace_bytes(B, S, T, D, flags)
        self.workspace = torch.empty(total, dtype=torch.uint8, device=dev)

        def carve(o, shape, dtype):
            n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            return self.workspace[o:o + n].view(dtype).view(*shape)

        self.hid, self.xn, self.y = carve(off[0], (B, N, D), BF16), carve(off[1], (B, N, D), BF16), carve(off[2], (B, N, 7 * D), BF16)
        self.q8 = carve(off[3], (B, N, 5 * D), torch.uint8) if self.fp8 else None
        self.q8_scale = carve(off[4], (B, N), torch.float32) if self.fp8 else None
        self.gemm_ws = carve(off[5], (gws.value // 4,), torch.float32)
        # RoPE tables: allocated ONCE per session and refreshed in place -- captured step graphs bake these pointers
        # into the kernel arguments, so a new ids layout for the same (B, S, T) must not move them
        self.cos = torch.empty(N, c.attention_head_dim, dtype=torch.float32, device=dev)
        self.sin = torch.empty(N, c.attention_head_dim, dtype=torch.float32, device=dev)
        self.rope_cs = torch.empty(N, c.attention_head_dim // 2, 2, dtype=torch.float32, device=dev)   # (cos_i, sin_i) pairs
        self._ids_key = None
        w = model.w

        w8 = model.w8   # {} unless enable_fp8() was called: e4m3 copies of the block linears + per-channel scales

        def lin(name):
            q = w8.get(name)
            return L.Linear(w[name + ".w"].data_ptr(), w[name + ".b"].data_ptr(), q[0].data_ptr() if q else None,
                            q[1].data_ptr() if q else None)

        bounds_d, bounds_s = model.attn_score_bounds()
        self._dbl = (L.DoubleBlock * max(1, c.num_layers))()
        for i in range(c.num_layers):
            b = self._dbl[i]
            for n in ("qkv_img", "qkv_txt", "out_img", "out_txt", "ff1_img", "ff2_img", "ff1_txt", "ff2_txt"):
                setattr(b, n, lin(f"d{i}.{n}"))
            for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                setattr(b, n, w[f"d{i}.{n}"].data_ptr())
            b.attn_score_bound = bounds_d[i]
        self._sgl = (L.SingleBlock * max(1, c.num_single_layers))()
        for j in range(c.num_single_layers):
            b = self._sgl[j]
            b.qkv_mlp, b.proj_out = lin(f"s{j}.qkv_mlp"), lin(f"s{j}.proj_out")
            b.norm_q, b.norm_k = w[f"s{j}.norm_q"].data_ptr(), w[f"s{j}.norm_k"].data_ptr()
            b.attn_score_bound = bounds_s[j]
        d = self.desc = L.DitDesc()
        d.D, d.H, d.in_channels, d.out_channels = D, c.num_attention_heads, c.in_channels, model.out_channels
        d.n_double, d.n_single = c.num_layers, c.num_single_layers
        d.B, d.S, d.T = B, S, T
        d.x_embedder, d.proj_out = lin("x_embedder"), lin("proj_out")
        d.dbl, d.sgl = self._dbl, self._sgl
        d.xin, d.ctx0 = self.xin.data_ptr(), self.ctx0.data_ptr()
        d.hid, d.xn, d.y, d.out = self.hid.data_ptr(), self.xn.data_ptr(), self.y.data_ptr(), self.out.data_ptr()
        d.first_block, d.last_block, d.flags = 0, -1, 0
        d.cos_tab, d.sin_tab = self.cos.data_ptr(), self.sin.data_ptr()
        d.rope_cs = self.rope_cs.data_ptr() if model.fuse_qk_norm_rope else None
        d.attn_score_bound = 0.0        # every block carries its own (ABI 6); no forward-wide promise on top
        if self.fp8:
            d.q8, d.q8_scale = self.q8.data_ptr(), self.q8_scale.data_ptr()
        # scratch for the split-K path of few-tile GEMMs (text stream, small batch x resolution): fp32 partials of at most
        # slices x tiles <= 256 tiles of 256 x 256, i.e. 64 MiB whatever the problem size (the library sizes the scratch at 128 MiB: the attention
        # launches of the blocks put their stream-K partials, 69.2 MB, in the same memory between GEMMs)
        d.gemm_workspace, d.gemm_workspace_bytes = self.gemm_ws.data_ptr(), self.gemm_ws.numel() * 4
        self.graphs = {}        # (sampler, ...) -> C-level step graph handle (tfx_dit_step_capture)
        self._gb = None
        self._gstream = None    # side stream the step graphs are captured / replayed on

    def __del__(self):
        try:
            for g in self.graphs.values():
                if g:
                    L.lib().tfx_graph_destroy(g)
        except Exception:
            pass

    def graph_stream(self) -> "torch.cuda.Stream":
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(device=self.model.device)
        return self._gstream

    def step_desc(self, gb, is_amo: bool, fuse_euler: bool = False) -> "L.StepDesc":
        """tfx_step_desc of one denoising step over the persistent graph buffers `gb` (graph_buffers()).  fuse_euler: sampler 2 --
        the step's dsigma row (the EULER_PAD columns behind each modulation row) gates proj_out's epilogue, the latents live in xin."""
        d = self.desc
        d.mod, d.mod_bstride = gb["mod_cur"].data_ptr(), gb["mod_cur"].stride(0)
        d.first_block, d.last_block, d.flags = 0, -1, (4 if self.fp8 else 0)
        d.euler_gate, d.euler_gate_bstride = (gb["mod_cur"].data_ptr() + 2 * self.model.mod_len, gb["mod_cur"].stride(0)) if fuse_euler else (None, 0)
        sd = L.StepDesc()
        sd.dit = d
        sd.mod_table, sd.mod_cur, sd.mod_step_elems = gb["mod_table"].data_ptr(), gb["mod_cur"].data_ptr(), gb["mod_cur"].numel()
        sd.step_ptr, sd.latents = gb["step"].data_ptr(), gb["lat"].data_ptr()
        sd.coef, sd.noise = gb["coef"].data_ptr(), gb["noise"].data_ptr() if is_amo else None
        sd.sampler = 1 if is_amo else 2 if fuse_euler else 0
        return sd

    def set_conditioning(self, prompt_embeds: torch.Tensor, txt_ids: torch.Tensor, img_ids: torch.Tensor) -> None:
        """context_embedder(prompt_embeds) -> ctx0; RoPE tables for cat(txt_ids, img_ids)."""
        m = self.model
        assert prompt_embeds.shape[:2] == (self.B, self.T)
        ops.gemm(prompt_embeds.contiguous(), m.w["context_embedder.w"], m.w["context_embedder.b"], out=self.ctx0)
        ids = torch.cat((txt_ids.detach().float().cpu(), img_ids.detach().float().cpu()), dim=0)
        key = (ids.shape, float(ids.sum()), float((ids * torch.arange(1, 4)).sum()))
        if key != self._ids_key:
            cos, sin = rope_tables(ids, m.config.axes_dims_rope)
            self.cos.copy_(cos)
            self.sin.copy_(sin)
            self.rope_cs.copy_(torch.stack((cos[:, 0::2], sin[:, 0::2]), dim=-1))
            self._ids_key = key

    def graph_buffers(self, n_steps: int, n_coef: int, lat_shape):
        """Persistent device buffers a captured step graph points at (modulation table of all steps, the current
        step's rows, scheduler coefficients, latent state, AMO noise, the int32 step cursor)."""
        gb = self._gb
        if gb is None or gb["mod_table"].shape[0] < n_steps or gb["coef"].numel() < n_coef:
            dev = self.model.device
            for g in self.graphs.values():
                if g:
                    L.lib().tfx_graph_destroy(g)
            self.graphs = {}
            W = self.model.mod_len + EULER_PAD     # every row carries the step's dsigma behind its modulation values
            gb = self._gb = dict(
                mod_table=torch.empty(n_steps, self.B, W, dtype=BF16, device=dev),
                mod_cur=torch.empty(self.B, W, dtype=BF16, device=dev),
                coef=torch.zeros(max(n_coef, 3 * n_steps), dtype=torch.float32, device=dev),
                lat=torch.empty(lat_shape, dtype=BF16, device=dev),
                noise=torch.zeros(lat_shape, dtype=torch.float32, device=dev),
                step=torch.zeros(1, dtype=torch.int32, device=dev))
        return gb

    def run(self, mod: torch.Tensor, first_block: int = 0, last_block: int = -1, flags: int = 0, euler: bool = False) -> torch.Tensor:
        """One transformer forward with modulation rows `mod` [B, mod_len] (a view into a table is fine).
        first_block/last_block/flags: partial runs for block-level tests (see tfx_dit_desc).  euler: `mod` rows are mod_len +
        EULER_PAD wide, the extra columns hold the step's bf16 dsigma: proj_out's epilogue applies the Euler update in place on
        xin[:, :, :out_channels] (tfx_dit_desc.euler_gate) and `out` is not written."""
        W = self.model.mod_len + (EULER_PAD if euler else 0)
        assert mod.shape == (self.B, W) and mod.stride(1) == 1 and self._ids_key is not None
        self._mod_keepalive = mod
        d = self.desc
        d.mod, d.mod_bstride = mod.data_ptr(), mod.stride(0)
        d.euler_gate, d.euler_gate_bstride = (mod.data_ptr() + 2 * self.model.mod_len, mod.stride(0)) if euler else (None, 0)
        d.first_block, d.last_block, d.flags = first_block, last_block, flags | (4 if self.fp8 else 0)
        L.check(L.lib().tfx_dit_forward(C.byref(d), ops._stream()), "dit_forward")
        return self.out
