"""GPU parity of the whole FluxTransformer2DModel.forward (C-ABI tfx_dit_forward) against the reference's golden
outputs (tests/golden/g3_model.safetensors: 2+2 layers, D=256, S=64, T=16) and the full-width block goldens.

Stated tolerance (north_star: latent MAE <= 1e-3 in bf16): the model output here has mean |x| ~ O(1); the engine must
be no further (relative MAE) from the fp32 reference than 1.25x, and from the bf16 reference than 1.5x, the distance
between the reference's own bf16 and fp32 runs (that distance, ~1e-2 on this 4-block model, is asserted too)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_oracle as fo
from oracle import pipeline_oracle as po

BF = torch.bfloat16
G3_CFG = fo.FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64,
                       pooled_projection_dim=32)


def build(cfg, seed):
    from textflux_amd.transformer import FluxTransformer2DModel
    m = FluxTransformer2DModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                               num_single_layers=cfg.num_single_layers, num_attention_heads=cfg.num_attention_heads,
                               joint_attention_dim=cfg.joint_attention_dim,
                               pooled_projection_dim=cfg.pooled_projection_dim, guidance_embeds=True)
    return m.load_state_dict(fo.seeded_state_dict(cfg, seed), device="cuda")


def rel_mae(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().mean() / b.abs().mean()).item()


def test_forward_matches_reference_goldens(golden):
    g = golden("g3_model")
    m = build(G3_CFG, 7)
    inp = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    out = m.forward(hidden_states=inp["hidden_states"].to(BF).cuda(),
                    encoder_hidden_states=inp["encoder_hidden_states"].to(BF).cuda(),
                    pooled_projections=inp["pooled_projections"].to(BF).cuda(), timestep=inp["timestep"].to(BF).cuda(),
                    img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=inp["guidance"].cuda(),
                    return_dict=False)[0]
    assert out.shape == g["out_bf16"].shape and torch.isfinite(out).all()
    ref_gap = rel_mae(g["out_bf16"], g["out_f32"])
    e_bf, e_f32 = rel_mae(out, g["out_bf16"]), rel_mae(out, g["out_f32"])
    print(f"rel MAE vs ref-bf16 {e_bf:.2e}, vs ref-f32 {e_f32:.2e}, ref-bf16 vs ref-f32 {ref_gap:.2e}")
    assert ref_gap < 2e-2
    assert e_f32 <= 1.25 * ref_gap and e_bf <= 1.5 * ref_gap


def test_forward_is_deterministic(golden):
    g = golden("g3_model")
    m = build(G3_CFG, 7)
    inp = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    kw = dict(hidden_states=inp["hidden_states"].to(BF).cuda(),
              encoder_hidden_states=inp["encoder_hidden_states"].to(BF).cuda(),
              pooled_projections=inp["pooled_projections"].to(BF).cuda(), timestep=inp["timestep"].to(BF).cuda(),
              img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=inp["guidance"].cuda(), return_dict=False)
    a = m.forward(**kw)[0]
    b = m.forward(**kw)[0]
    assert torch.equal(a, b)  # no atomics / split-K anywhere: bit-identical reruns


def test_full_width_blocks_match_reference_goldens(golden):
    """One double + one single block at the real width (D = 3072, 24 heads) against the reference block outputs,
    driven through tfx_dit_forward's block range (first_block/last_block)."""
    g = golden("g2_blocks")
    heads, S, T, seed, h2, w2 = [int(v) for v in g["d3072.meta"]]
    D = heads * 128
    cfg = fo.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=heads, joint_attention_dim=64,
                        pooled_projection_dim=32)
    sd = fo.seeded_state_dict(cfg, seed)
    m = build(cfg, seed)

    def rnd(shape, s):
        return torch.randn(shape, generator=torch.Generator().manual_seed(s))

    hidden, enc, temb = rnd((2, S, D), seed + 1), rnd((2, T, D), seed + 2), rnd((2, D), seed + 3)
    ses = m.session(2, S, T)
    ids_img, ids_txt = po.latent_image_ids(h2, w2), torch.zeros(T, 3)
    ses.set_conditioning(torch.zeros(2, T, 64, dtype=BF, device="cuda"), ids_txt, ids_img)
    mod = m.modulation(temb.to(BF).cuda())
    # double block alone: preload the joint stream [text | image], run block 0, no embedders / no final projection
    ses.hid[:, :T].copy_(enc.to(BF))
    ses.hid[:, T:].copy_(hidden.to(BF))
    ses.run(mod, first_block=0, last_block=1, flags=3)
    got = ses.hid.clone()
    e_enc, e_hid = rel_mae(got[:, :T], g["d3072.double.enc_out"]), rel_mae(got[:, T:], g["d3072.double.hidden_out"])
    # single block alone on the SAME joint input the reference block saw
    ses.hid[:, :T].copy_(enc.to(BF))
    ses.hid[:, T:].copy_(hidden.to(BF))
    ses.run(mod, first_block=1, last_block=2, flags=3)
    e_sgl = rel_mae(ses.hid, g["d3072.single.out"])
    print(f"full-width rel MAE: double enc {e_enc:.2e} hidden {e_hid:.2e}; single {e_sgl:.2e}")
    assert max(e_enc, e_hid, e_sgl) < 1e-2


# ----------------------------------------------------------------------------- fp8 linears (BASELINE config 5)
def _g3_inputs(golden):
    g = golden("g3_model")
    return g, {k[3:]: v for k, v in g.items() if k.startswith("in.")}


def test_fp8_forward_matches_fp8_oracle(golden):
    """enable_fp8(): e4m3 weights (per-channel scale) x e4m3 activations (per-token scale) on the fp8 MFMA.  Checked
    against the oracle running the same quantisation scheme (oracle.flux_oracle.fp8_block_linears) in bf16, with the
    bf16 criterion of the test above: no further from it than 1.5x the reference's own bf16-vs-fp32 distance; the cost of
    fp8 against the bf16 engine is printed (and bounded loosely: it is a property of the scheme, not of the kernels)."""
    g, inp = _g3_inputs(golden)
    m = build(G3_CFG, 7)
    kw = dict(hidden_states=inp["hidden_states"].to(BF).cuda(), encoder_hidden_states=inp["encoder_hidden_states"].to(BF).cuda(),
              pooled_projections=inp["pooled_projections"].to(BF).cuda(), timestep=inp["timestep"].to(BF).cuda(),
              img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=inp["guidance"].cuda(), return_dict=False)
    out_bf16 = m.forward(**kw)[0].clone()
    m.enable_fp8()
    assert len(m.w8) == 2 * 8 + 2 * 2
    out_fp8 = m.forward(**kw)[0].clone()
    assert torch.isfinite(out_fp8).all() and not torch.equal(out_fp8, out_bf16)
    sd = {k: v.to(BF) for k, v in fo.seeded_state_dict(G3_CFG, 7).items()}
    with fo.fp8_block_linears():
        ref = fo.transformer_forward(sd, G3_CFG, inp["hidden_states"].to(BF), inp["encoder_hidden_states"].to(BF),
                                     inp["pooled_projections"].to(BF), inp["timestep"].to(BF), inp["img_ids"], inp["txt_ids"],
                                     inp["guidance"])
    ref_gap = rel_mae(g["out_bf16"], g["out_f32"])
    e = rel_mae(out_fp8, ref)
    cost = rel_mae(out_fp8, out_bf16)
    print(f"fp8 engine vs fp8 oracle rel MAE {e:.2e} (bf16 ref gap {ref_gap:.2e}); fp8 vs bf16 engine {cost:.2e}")
    assert e <= 1.5 * ref_gap
    assert cost <= 0.15
    m.enable_fp8(False)
    assert torch.equal(m.forward(**kw)[0], out_bf16)


def test_fp8_full_width_blocks_match_fp8_oracle(golden):
    """One double + one single block at the real width (D = 3072: the fp8 GEMMs at K = 3072 / 12288 / 15360, split-K on the
    few-tile ones) against the oracle's own blocks under fp8_block_linears (bf16), same inputs as the bf16 full-width test.
    Quantisation is discontinuous: a one-ulp bf16 difference upstream (LayerNorm summation order, say) moves a row's scale
    or flips e4m3 codes (12 % steps), so engine and oracle decorrelate at the level of the fp8 noise itself.  Bound: the
    engine is closer to the fp8 oracle than fp8 is to the reference's bf16 blocks (the cost of the scheme, also bounded),
    and within 3e-2 relative MAE."""
    g = golden("g2_blocks")
    heads, S, T, seed, h2, w2 = [int(v) for v in g["d3072.meta"]]
    D = heads * 128
    cfg = fo.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=heads, joint_attention_dim=64,
                        pooled_projection_dim=32)
    sd = {k: v.to(BF) for k, v in fo.seeded_state_dict(cfg, seed).items()}
    m = build(cfg, seed).enable_fp8()

    def rnd(shape, s):
        return torch.randn(shape, generator=torch.Generator().manual_seed(s))

    hidden, enc, temb = rnd((2, S, D), seed + 1).to(BF), rnd((2, T, D), seed + 2).to(BF), rnd((2, D), seed + 3).to(BF)
    ses = m.session(2, S, T)
    ids_img, ids_txt = po.latent_image_ids(h2, w2), torch.zeros(T, 3)
    ses.set_conditioning(torch.zeros(2, T, 64, dtype=BF, device="cuda"), ids_txt, ids_img)
    mod = m.modulation(temb.cuda())
    cos, sin = fo.flux_pos_embed(torch.cat([ids_txt, ids_img]))
    with fo.fp8_block_linears():
        r_enc, r_hid = fo.double_block(sd, "transformer_blocks.0", heads, hidden, enc, temb, cos, sin)
        r_sgl = fo.single_block(sd, "single_transformer_blocks.0", heads, torch.cat([enc, hidden], 1), temb, cos, sin)
    ses.hid[:, :T].copy_(enc)
    ses.hid[:, T:].copy_(hidden)
    ses.run(mod, first_block=0, last_block=1, flags=3)
    got = ses.hid.clone()
    e_enc, e_hid = rel_mae(got[:, :T], r_enc), rel_mae(got[:, T:], r_hid)
    c_hid = rel_mae(got[:, T:], g["d3072.double.hidden_out"])
    ses.hid[:, :T].copy_(enc)
    ses.hid[:, T:].copy_(hidden)
    ses.run(mod, first_block=1, last_block=2, flags=3)
    e_sgl, c_sgl = rel_mae(ses.hid, r_sgl), rel_mae(ses.hid, g["d3072.single.out"])
    print(f"fp8 full-width rel MAE vs fp8 oracle: double enc {e_enc:.2e} hidden {e_hid:.2e}; single {e_sgl:.2e}; "
          f"vs the reference's bf16 blocks: double {c_hid:.2e} single {c_sgl:.2e}")
    assert max(e_enc, e_hid, e_sgl) < 3e-2
    assert e_hid < c_hid and e_sgl < c_sgl
    assert max(c_hid, c_sgl) < 6e-2


def test_3d_ids_and_return_dict_follow_the_reference(golden):
    """The reference's model tests (diffusers/tests/models/transformers/test_models_transformer_flux.py:87-113): `txt_ids` / `img_ids` with
    a leading batch dimension (the deprecated 3-D form, transformer_flux.py:1100-1111 takes sample 0) give the 2-D result; and the
    dict / tuple equivalence of PipelineTesterMixin (`return_dict=True` -> an object with `.sample`, False -> a 1-tuple)."""
    g = golden("g3_model")
    m = build(G3_CFG, 7)
    inp = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    kw = dict(hidden_states=inp["hidden_states"].to(BF).cuda(), encoder_hidden_states=inp["encoder_hidden_states"].to(BF).cuda(),
              pooled_projections=inp["pooled_projections"].to(BF).cuda(), timestep=inp["timestep"].to(BF).cuda(),
              guidance=inp["guidance"].cuda())
    B = inp["hidden_states"].shape[0]
    two_d = m.forward(img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], return_dict=False, **kw)
    three_d = m.forward(img_ids=inp["img_ids"][None].repeat(B, 1, 1), txt_ids=inp["txt_ids"][None].repeat(B, 1, 1), return_dict=False, **kw)
    assert isinstance(two_d, tuple) and len(two_d) == 1 and torch.equal(two_d[0], three_d[0])
    obj = m.forward(img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], return_dict=True, **kw)
    assert torch.equal(obj.sample, two_d[0])


def test_fp8_fused_qk_norm_rope_epilogue_matches_the_separate_pass():
    """the e4m3 projections carry the fused q / k RMSNorm + RoPE epilogue since round 5 (`fp8_fuse_qkn`, default 1); the
    header claims "same rounding points either way" (dequantise, bias, bf16 round, RMSNorm, RoPE).  A/B on one double and one single block
    at the production width and a token count at which the launches are eligible (>= one tile per CU): the k columns -- normalised and
    rotated, not overwritten by attention -- agree to one bf16 step of their rotation pair on all but a sliver of elements (another fp32
    summation order of the 128 squares), the v columns bit for bit up to the compiler's FMA contraction (about one element in a million one bf16 step apart), both modes deterministic, and the knob really switches paths."""
    from textflux_amd import ops
    from textflux_amd.transformer import FluxTransformer2DModel
    D = 3072
    m = FluxTransformer2DModel(in_channels=384, out_channels=64, num_layers=1, num_single_layers=1,
                               guidance_embeds=True).init_random_(seed=9, device="cuda").enable_fp8()
    g = torch.Generator(device="cuda").manual_seed(4)
    B, S, T = 2, 4096, 512
    temb = torch.randn(B, D, generator=g, device="cuda").to(BF)
    hid0 = torch.randn(B, S + T, D, generator=g, device="cuda").to(BF)
    ids = torch.zeros(S, 3)
    ids[:, 1] = torch.arange(S) // 64
    ids[:, 2] = torch.arange(S) % 64
    ses = m.session(B, S, T)
    ses.set_conditioning(torch.zeros(B, T, 4096, dtype=BF, device="cuda"), torch.zeros(T, 3), ids)
    assert ses.desc.rope_cs is not None
    mod = m.modulation(temb)
    ks, vs = {}, {}
    try:
        for fused in (1, 0, 1, 0):
            ops.set_option("fp8_fuse_qkn", fused)
            for blk in (0, 1):
                ses.hid.copy_(hid0)
                ses.run(mod, first_block=blk, last_block=blk + 1, flags=3 | 4)
                k_, v_ = ses.y[:, :, :D].clone(), ses.y[:, :, D:2 * D].clone()
                if (fused, blk) in ks:          # second visit of a mode: the same launches must give the same bits
                    for name, a_, b_ in (("k", ks[(fused, blk)], k_), ("v", vs[(fused, blk)], v_)):
                        nz = (a_ != b_).nonzero()
                        assert nz.numel() == 0, (f"fp8_fuse_qkn={fused} block {blk}: {name} columns differ between two identical runs at", nz[:8].tolist())
                ks[(fused, blk)], vs[(fused, blk)] = k_, v_
    finally:
        ops.set_option("fp8_fuse_qkn", 1)
    for blk in (0, 1):
        # the v columns take the plain bias path in both kernels -- two instantiations of one template, in which hipcc contracts
        # (acc * a_scale) * w_scale + bias into an FMA or not as it sees fit: a last-fp32-bit difference before the bf16 rounding, i.e. about
        # one element in a million one bf16 step apart (measured: 30 of 28.3 M), deterministic within a mode (asserted above)
        va, vb = vs[(1, blk)].float(), vs[(0, blk)].float()
        nd = int((va != vb).sum())
        print(f"fp8 block {blk}: v columns fused vs separate: {nd} of {va.numel()} elements differ")
        assert nd <= 1e-5 * va.numel() and ((va - vb).abs() <= 2 ** -7 * vb.abs() + 1e-6).all()
        a, b = ks[(0, blk)].float(), ks[(1, blk)].float()
        assert torch.isfinite(b).all()
        diff = (a - b).abs()
        pair = (a.view(*a.shape[:-1], -1, 2) ** 2).sum(-1).sqrt().repeat_interleave(2, dim=-1)
        assert (diff <= 2 ** -6 * pair + 1e-6).all(), (blk, (diff / (2 ** -6 * pair + 1e-6)).max().item())
        frac = (diff > 0).float().mean().item()
        print(f"fp8 block {blk}: fused vs separate q/k norm + RoPE: {frac:.2e} of the k elements differ, all within one bf16 step of their pair")
        assert 0 < frac < 2e-2          # > 0: the knob switched paths (the separate pass sums the squares in another order)
