"""The HIP text encoders (textflux_amd/text_encoders.py: T5 v1.1 encoder, CLIP text model) against the CPU oracle
(oracle/text_oracle.py, pinned against `transformers` in tests/test_text_oracle.py): the tiny golden configurations, and
T5-XXL / CLIP-L layer shapes at reduced depth (d_model 4096, 64 heads, d_ff 10240, 512 tokens; 768 wide, 12 heads, 77 tokens)."""
import pytest
import torch

from oracle import text_oracle as to

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def sub(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def rel_mae(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().mean() / b.abs().mean()).item()


@pytest.mark.parametrize("N,H,causal,bias", [(40, 2, False, True), (77, 12, True, False), (512, 4, False, True), (129, 3, True, True)])
def test_attention64_matches_fp32_attention(N, H, causal, bias):
    from textflux_amd import ops
    g = torch.Generator().manual_seed(N)
    B = 2
    q, k, v = (torch.randn(B, N, H * 64, generator=g).to(BF) for _ in range(3))
    tab = torch.randn(H, 2 * N - 1, generator=g) if bias else None
    scale = 1.0 if bias else 0.125
    qh, kh, vh = (t.float().view(B, N, H, 64).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if bias:
        idx = torch.arange(N)[None, :] - torch.arange(N)[:, None] + N - 1          # key - query + N - 1
        s = s + tab[:, idx][None]
    if causal:
        s = s + torch.full((N, N), float("-inf")).triu(1)
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, N, H * 64)
    got = ops.attention64(q.cuda(), k.cuda(), v.cuda(), scale, rel_bias=tab.cuda() if bias else None, causal=causal)
    err = (got.float().cpu() - ref).abs()
    assert err.max().item() < 2e-2 and err.mean().item() < 1.5e-3, (err.max().item(), err.mean().item())


def test_t5_tiny_matches_oracle_and_transformers_golden(golden):
    from textflux_amd.text_encoders import T5EncoderModel
    g = golden("g10_text")
    sd = sub(g, "t5.sd.")
    cfg = dict(d_model=64, d_kv=64, num_heads=2, d_ff=128, num_layers=2, vocab_size=100, feed_forward_proj="gated-gelu")
    m = T5EncoderModel(cfg).load_state_dict(sd, device="cuda")
    out = m(g["t5.ids"].cuda(), output_hidden_states=False)[0]
    ocfg = to.T5Cfg(d_model=64, d_kv=64, num_heads=2, d_ff=128, num_layers=2)
    ref_bf = to.t5_encode(to.t5_state_dict_as_loaded(sd), ocfg, g["t5.ids"])        # the reference's bf16 load semantics
    assert out.shape == g["t5.out"].shape and out.dtype == BF
    e_bf, e_f32, gap = rel_mae(out, ref_bf), rel_mae(out, g["t5.out"]), rel_mae(ref_bf, g["t5.out"])
    print(f"T5 tiny: engine vs bf16-loaded oracle {e_bf:.2e}, vs transformers fp32 {e_f32:.2e}; oracle bf16 vs fp32 {gap:.2e}")
    assert e_f32 <= 1.5 * gap + 1e-3 and e_bf <= 1.5 * gap + 1e-3
    # transformers' own bf16 run (all-bf16 load, bf16 residual stream): the engine rounds the stream at the same points
    assert torch.equal(ref_bf, g["t5.out_bf16"])
    e_gold = rel_mae(out, g["t5.out_bf16"])
    print(f"T5 tiny: engine vs transformers bf16 golden {e_gold:.2e}")
    assert e_gold <= gap + 1e-3


def test_clip_tiny_matches_oracle_and_transformers_golden(golden):
    from textflux_amd.text_encoders import CLIPTextModel
    g = golden("g10_text")
    sd = sub(g, "clip.sd.")
    cfg = dict(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=2)
    m = CLIPTextModel(cfg).load_state_dict(sd, device="cuda")
    r = m(g["clip.ids"].cuda(), output_hidden_states=False)
    ocfg = to.ClipCfg(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2)
    last_bf, pooled_bf = to.clip_encode({k: v.to(BF) for k, v in sd.items()}, ocfg, g["clip.ids"])
    gap = rel_mae(pooled_bf, g["clip.pooled"])
    e = rel_mae(r.pooler_output, g["clip.pooled"])
    print(f"CLIP tiny pooled: engine vs transformers fp32 {e:.2e}; oracle bf16 vs fp32 {gap:.2e}")
    assert r.pooler_output.shape == (2, 128) and e <= 1.5 * gap + 1e-3
    assert rel_mae(r.last_hidden_state, g["clip.last"]) <= 1.5 * rel_mae(last_bf, g["clip.last"]) + 1e-3
    assert rel_mae(r.pooler_output, g["clip.pooled_bf16"]) <= gap + 1e-3


def test_clip_layernorm_any_gamma(golden):
    """LayerNorm weights far from 1 -- in (0, 0.5), negative, large -- load and compute (round 2 re-parameterised gamma as
    1 + scale and refused what that could not represent exactly; real CLIP-L checkpoints hold such values)."""
    from textflux_amd import ops
    from textflux_amd.text_encoders import CLIPTextModel
    g = golden("g10_text")
    sd = {k: v.clone() for k, v in sub(g, "clip.sd.").items()}
    gen = torch.Generator().manual_seed(3)
    for k in sd:
        if "layer_norm" in k and k.endswith(".weight"):
            n = sd[k].numel()
            w = torch.empty(n)
            w[0::4] = torch.rand(n // 4, generator=gen) * 0.5                 # (0, 0.5)
            w[1::4] = -torch.rand(n // 4, generator=gen) * 1.5                # negative
            w[2::4] = 2.0 + 3.0 * torch.rand(n // 4, generator=gen)           # > 2
            w[3::4] = 1.0 + 0.01 * torch.randn(n // 4, generator=gen)
            sd[k] = w
    cfg = dict(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=2)
    r = CLIPTextModel(cfg).load_state_dict(sd, device="cuda")(g["clip.ids"].cuda())
    ocfg = to.ClipCfg(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2)
    last32, pooled32 = to.clip_encode(sd, ocfg, g["clip.ids"])
    last_bf, pooled_bf = to.clip_encode({k: v.to(BF) for k, v in sd.items()}, ocfg, g["clip.ids"])
    e, gap = rel_mae(r.last_hidden_state, last32), rel_mae(last_bf, last32)
    print(f"CLIP with wild LayerNorm weights: engine vs fp32 oracle {e:.2e}; bf16 oracle vs fp32 {gap:.2e}")
    assert torch.isfinite(r.last_hidden_state.float()).all() and e <= 1.5 * gap + 1e-3
    # the kernel alone: one rounding, as F.layer_norm on bf16 tensors
    x = torch.randn(5, 77, 128, generator=gen).to(BF)
    ga, be = sd["text_model.final_layer_norm.weight"].to(BF), sd["text_model.final_layer_norm.bias"].to(BF)
    ref = torch.nn.functional.layer_norm(x, (128,), ga, be, 1e-5)
    got = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5).cpu()
    d = (got.float() - ref.float()).abs()
    assert (d <= 2 ** -7 * ref.float().abs() + 1e-6).all() and (d > 0).float().mean().item() < 0.02


def _seeded(shapes, seed, std=0.02):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in shapes.items():
        r = torch.randn(s, generator=g)
        sd[k] = (1.0 + 0.1 * r) if ("layer_norm" in k and k.endswith("weight")) else std * r
    return sd


def test_t5_xxl_layer_shapes_match_oracle():
    """Two layers at the T5-XXL geometry (d_model 4096, 64 heads x 64, d_ff 10240), 512 tokens: the production GEMM /
    attention shapes, against the oracle under the reference's bf16 load semantics (every weight and the residual stream in bf16)."""
    from textflux_amd.text_encoders import T5EncoderModel
    D, Hh, dff, L, V, T = 4096, 64, 10240, 2, 1000, 512
    shapes = {"shared.weight": (V, D), "encoder.final_layer_norm.weight": (D,),
              "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": (32, Hh)}
    for i in range(L):
        p = f"encoder.block.{i}.layer."
        for n in "qkv":
            shapes[p + f"0.SelfAttention.{n}.weight"] = (Hh * 64, D)
        shapes[p + "0.SelfAttention.o.weight"] = (D, Hh * 64)
        shapes[p + "0.layer_norm.weight"] = shapes[p + "1.layer_norm.weight"] = (D,)
        shapes[p + "1.DenseReluDense.wi_0.weight"] = shapes[p + "1.DenseReluDense.wi_1.weight"] = (dff, D)
        shapes[p + "1.DenseReluDense.wo.weight"] = (D, dff)
    sd = _seeded(shapes, 5)
    sd["shared.weight"] = sd["shared.weight"] * 50          # unit-scale embeddings
    sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"] *= 50
    ids = torch.randint(0, V, (1, T), generator=torch.Generator().manual_seed(6))
    cfg = dict(d_model=D, d_kv=64, num_heads=Hh, d_ff=dff, num_layers=L, vocab_size=V, feed_forward_proj="gated-gelu")
    out = T5EncoderModel(cfg).load_state_dict(sd, device="cuda")(ids.cuda())[0]
    ocfg = to.T5Cfg(num_layers=L)
    ref = to.t5_encode(to.t5_state_dict_as_loaded(sd), ocfg, ids)
    ref32 = to.t5_encode(sd, ocfg, ids)
    e, gap = rel_mae(out, ref32), rel_mae(ref, ref32)
    print(f"T5-XXL shapes (2 layers, 512 tokens): engine vs fp32 oracle {e:.2e}; bf16-loaded oracle vs fp32 oracle {gap:.2e}")
    assert e <= 1.5 * gap + 1e-3


def test_clip_l_layer_shapes_match_oracle():
    from textflux_amd.text_encoders import CLIPTextModel
    D, Hh, I, L, V, T = 768, 12, 3072, 2, 1000, 77
    shapes = {"text_model.embeddings.token_embedding.weight": (V, D), "text_model.embeddings.position_embedding.weight": (T, D),
              "text_model.final_layer_norm.weight": (D,), "text_model.final_layer_norm.bias": (D,)}
    for i in range(L):
        l = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            shapes[l + f"self_attn.{n}.weight"], shapes[l + f"self_attn.{n}.bias"] = (D, D), (D,)
        for n in ("layer_norm1", "layer_norm2"):
            shapes[l + n + ".weight"], shapes[l + n + ".bias"] = (D,), (D,)
        shapes[l + "mlp.fc1.weight"], shapes[l + "mlp.fc1.bias"] = (I, D), (I,)
        shapes[l + "mlp.fc2.weight"], shapes[l + "mlp.fc2.bias"] = (D, I), (D,)
    sd = {k: v.to(BF).float() for k, v in _seeded(shapes, 7).items()}
    sd["text_model.embeddings.token_embedding.weight"] *= 50
    ids = torch.randint(0, V - 1, (3, T), generator=torch.Generator().manual_seed(8))
    ids[:, 40] = V - 1
    cfg = dict(vocab_size=V, hidden_size=D, intermediate_size=I, num_hidden_layers=L, num_attention_heads=Hh,
               max_position_embeddings=T, hidden_act="quick_gelu", eos_token_id=2)
    r = CLIPTextModel(cfg).load_state_dict(sd, device="cuda")(ids.cuda())
    ocfg = to.ClipCfg(num_hidden_layers=L)
    last32, pooled32 = to.clip_encode(sd, ocfg, ids)
    last_bf, pooled_bf = to.clip_encode({k: v.to(BF) for k, v in sd.items()}, ocfg, ids)
    e, gap = rel_mae(r.pooler_output, pooled32), rel_mae(pooled_bf, pooled32)
    print(f"CLIP-L shapes (2 layers): pooled engine vs fp32 oracle {e:.2e}; bf16 oracle vs fp32 oracle {gap:.2e}")
    assert e <= 1.5 * gap + 1e-3 and torch.equal(r.pooler_output, r.last_hidden_state[:, 40])
