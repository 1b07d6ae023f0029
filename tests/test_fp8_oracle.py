"""CPU checks of the fp8 (e4m3) emulation the oracle uses to judge the fp8 engine (BASELINE config 5; the reference has no
fp8 path, so this pins the scheme's definition: per-row absmax / 448 scales, OCP e4m3 round-to-nearest-even)."""
import torch

from oracle import flux_oracle as fo

CFG = fo.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)


def test_quantize_rows_e4m3_definition():
    x = torch.randn(5, 7, 256, generator=torch.Generator().manual_seed(0)) * torch.tensor([0.01, 1.0, 30.0, 1e3, 1e-4]).view(5, 1, 1)
    x[1, 2] = 0
    q, s = fo.quantize_rows_e4m3(x)
    assert s.shape == (5, 7, 1) and q.shape == x.shape
    assert torch.equal(s[1, 2], torch.ones(1)) and q[1, 2].abs().max() == 0
    nz = torch.ones(5, 7, dtype=torch.bool)
    nz[1, 2] = False
    assert torch.equal(s[nz].squeeze(-1), x.abs().amax(-1)[nz] / 448.0)
    assert q.abs().max() <= 448 and torch.equal(q.abs().amax(-1)[nz], torch.full((34,), 448.0))   # the row maximum maps to 448
    # e4m3 has 3 mantissa bits: relative error <= 2^-4 for values in the normal range (|q| >= 2^-6)
    deq = q * s
    normal = q.abs() >= 2 ** -6
    assert ((deq - x).abs()[normal] <= x.abs()[normal] * 2 ** -4 * 1.0001).all()
    # every code is a representable e4m3 value: casting again changes nothing
    assert torch.equal(q.to(torch.float8_e4m3fn).float(), q)


def test_fp8_context_touches_only_block_linears():
    assert fo._is_block_linear("transformer_blocks.0.attn.to_q") and fo._is_block_linear("single_transformer_blocks.3.proj_out")
    assert fo._is_block_linear("transformer_blocks.1.ff_context.net.2")
    for name in ("transformer_blocks.0.norm1.linear", "single_transformer_blocks.0.norm.linear", "x_embedder", "context_embedder",
                 "proj_out", "norm_out.linear", "time_text_embed.timestep_embedder.linear_1"):
        assert not fo._is_block_linear(name)
    sd = fo.seeded_state_dict(CFG, 3)
    g = torch.Generator().manual_seed(1)
    kw = dict(hidden_states=torch.randn(1, 16, 384, generator=g), encoder_hidden_states=torch.randn(1, 8, 64, generator=g),
              pooled_projections=torch.randn(1, 32, generator=g), timestep=torch.tensor([0.7]),
              img_ids=torch.zeros(16, 3), txt_ids=torch.zeros(8, 3), guidance=torch.tensor([30.0]))
    ref = fo.transformer_forward(sd, CFG, **kw)
    with fo.fp8_block_linears():
        q = fo.transformer_forward(sd, CFG, **kw)
    again = fo.transformer_forward(sd, CFG, **kw)
    assert torch.equal(ref, again)                       # the context manager restores the exact path
    rel = ((q - ref).abs().mean() / ref.abs().mean()).item()
    assert 1e-5 < rel < 0.1, rel                          # fp8 changes the result a little, and only a little
