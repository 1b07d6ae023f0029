"""GPU parity of the VAE-side HIP kernels (implicit-GEMM 3x3 convolution, GroupNorm+SiLU on NHWC) against
torch.nn.functional on the CPU in fp32 (the ops the CPU oracle oracle/vae_oracle.py is made of)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from textflux_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, max_rel=1.5e-2, mae_rel=3e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    assert err.max().item() <= max_rel * ref.abs().max().item() + 1e-5, (err.max().item(), ref.abs().max().item())
    assert err.mean().item() <= mae_rel * ref.abs().mean().item() + 1e-6


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up,pad_lo,with_res", [
    (2, 12, 20, 64, 64, 1, 1, 1, False),      # plain 3x3, pad 1
    (1, 9, 7, 128, 72, 1, 1, 1, True),        # ragged sizes, residual add, Cout not a tile multiple
    (2, 8, 6, 64, 128, 1, 2, 1, False),       # nearest 2x upsample folded into the gather (Upsample2D)
    (1, 16, 12, 64, 64, 2, 1, 0, False),      # Downsample2D: pad (0,1,0,1), stride 2
    (1, 40, 36, 256, 256, 1, 1, 1, True),     # several K tiles per tap, > 1 row tile
])
def test_conv3x3_nhwc(ops, variant, B, H, W, Cin, Cout, stride, up, pad_lo, with_res):
    x = rnd((B, Cin, H, W), 1).to(BF)
    w = rnd((Cout, Cin, 3, 3), 2, 0.05).to(BF)
    b = rnd((Cout,), 3).to(BF)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up == 2 else x.float()
    if stride == 2:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), b.float(), stride=2, padding=0)
    else:
        ref = F.conv2d(xin, w.float(), b.float(), stride=1, padding=1)
    res = rnd(ref.shape, 4).to(BF) if with_res else None
    if with_res:
        ref = res.float() + ref.to(BF).float()
    got = ops.conv3x3_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda(),
                           stride=stride, up=up, pad_lo=pad_lo,
                           res=res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None, variant=variant)
    assert got.shape == (B, ref.shape[2], ref.shape[3], Cout)
    close(got.permute(0, 3, 1, 2), ref.to(BF))


@pytest.mark.parametrize("B,H,W,Cin,Cout,with_res", [(2, 12, 20, 64, 128, False), (1, 9, 14, 128, 128, True), (1, 40, 36, 256, 128, True),
                                                     (2, 7, 6, 64, 72, False)])
def test_conv3x3_pixel_pair_form(ops, B, H, W, Cin, Cout, with_res):
    """Round 4: layers with <= 128 output channels as one GEMM row per PAIR of horizontally adjacent pixels (N = 2 Cout fills the
    256-column tile; K over the 3 x 4 taps the pair touches, zero weights where a pixel does not use a column) -- against fp32
    F.conv2d, and bit for bit against the one-pixel-per-row form."""
    x = rnd((B, Cin, H, W), 11).to(BF)
    w = rnd((Cout, Cin, 3, 3), 12, 0.05).to(BF)
    b = rnd((Cout,), 13).to(BF)
    ref = F.conv2d(x.float(), w.float(), b.float(), stride=1, padding=1)
    res = rnd(ref.shape, 14).to(BF) if with_res else None
    if with_res:
        ref = res.float() + ref.to(BF).float()
    xn, wk = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda()
    rn = res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None
    wp, bp = ops.pair_conv_weights(wk, b.cuda())
    assert wp.shape == (2 * Cout, 3, 4, Cin) and bp.shape == (2 * Cout,)
    got = ops.conv3x3_pair_nhwc(xn, wp, bp, res=rn)
    assert got.shape == (B, H, W, Cout)
    close(got.permute(0, 3, 1, 2), ref.to(BF))
    # the pair form visits the same (tap, channel) products in the same order -- the extra taps are exact zeros -- so it is
    # bit-identical to the one-pixel-per-row form, not merely close
    assert torch.equal(got, ops.conv3x3_nhwc(xn, wk, b.cuda(), res=rn))
    with pytest.raises(RuntimeError, match="W must be even"):
        ops.conv3x3_pair_nhwc(xn[:, :, :W - 1].contiguous(), wp, bp)


@pytest.mark.parametrize("C,groups,HW,silu", [(128, 32, 48 * 40, True), (256, 32, 1500, True), (512, 32, 33 * 31, False),
                                               (64, 4, 2100, True)])
def test_groupnorm_silu_nhwc(ops, C, groups, HW, silu):
    B = 2
    x = (rnd((B, C, HW), 5) * 1.5 + 0.3).to(BF)
    ga, be = (1 + 0.2 * rnd((C,), 6)).to(BF), (0.1 * rnd((C,), 7)).to(BF)
    ref = F.group_norm(x.float(), groups, ga.float(), be.float(), eps=1e-6).to(BF).float()
    if silu:
        ref = F.silu(ref)
    got = ops.groupnorm_nhwc(x.transpose(1, 2).contiguous().cuda(), ga.cuda(), be.cuda(), groups, silu=silu)
    close(got.transpose(1, 2), ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 20, 24, 3, 64), (1, 9, 13, 16, 128), (1, 33, 17, 3, 128), (2, 8, 8, 32, 64)])
def test_conv3x3_narrow_input(ops, variant, B, H, W, Cin, Cout):
    """conv_in of the two VAE ends: Cin padded to 8 / 16 / 32 channels, K = 9 * Cin_p padded with zero weights to a multiple
    of 64 (a K-tile spans several taps; taps >= 9 read the zero page)."""
    x = rnd((B, Cin, H, W), 21).to(BF)
    w = rnd((Cout, Cin, 3, 3), 22, 0.2).to(BF)
    b = rnd((Cout,), 23).to(BF)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    cp = 8 if Cin <= 8 else 16 if Cin <= 16 else 32
    xp = torch.zeros(B, H, W, cp, dtype=BF)
    xp[..., :Cin] = x.permute(0, 2, 3, 1)
    wp = torch.zeros(Cout, 3, 3, cp, dtype=BF)
    wp[..., :Cin] = w.permute(0, 2, 3, 1)
    kp = (9 * cp + 63) // 64 * 64
    wk = torch.zeros(Cout, kp, dtype=BF)
    wk[:, :9 * cp] = wp.reshape(Cout, 9 * cp)
    got = ops.conv3x3_nhwc(xp.cuda(), wk.cuda(), b.cuda(), variant=variant)
    close(got.permute(0, 3, 1, 2), ref.to(BF))


def test_vae_nhwc_path_matches_oracle():
    """Whole encoder / decoder on the HIP NHWC path against the fp32 CPU oracle -- including both narrow conv_in layers
    and the mid-block attention (score GEMM -> row softmax -> P v GEMM); the NCHW tensor interface and the NHWC entry
    points the pipeline uses must agree bit for bit."""
    from oracle import vae_oracle as vo
    from textflux_amd import ops as o
    from textflux_amd.vae import AutoencoderKL
    kw = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)
    cfg = vo.VaeConfig(**kw)
    sd = vo.seeded_state_dict(cfg, 321)
    vae = AutoencoderKL(**kw).load_state_dict(sd, device="cuda")
    x = rnd((2, 3, 48, 40), 11).clamp(-1, 1)
    z = rnd((2, 16, 12, 10), 12)
    mean, std = vo.encode_moments(x, sd, cfg)
    post = vae.encode(x.to(BF).cuda()).latent_dist
    dec_ref = vo.decoder(z, sd, cfg)
    dec = vae.decode(z.to(BF).cuda(), return_dict=False)[0]
    assert dec.shape == dec_ref.shape and post.mean.shape == mean.shape
    close(post.mean, mean, max_rel=5e-2, mae_rel=1.5e-2)
    close(post.std, std, max_rel=5e-2, mae_rel=1.5e-2)
    close(dec, dec_ref, max_rel=5e-2, mae_rel=1.5e-2)
    mom = vae.encode_moments_nhwc(o.prep_image(x.to(BF).cuda(), None))
    assert torch.equal(mom.permute(0, 3, 1, 2)[:, :16], post.mean)
    img = vae.decode_nhwc(z.to(BF).permute(0, 2, 3, 1).contiguous().cuda())
    assert torch.equal(img[..., :3].permute(0, 3, 1, 2), dec)


def test_vae_slicing_is_bit_identical_to_the_batched_path():
    """enable_slicing() (reference: autoencoder_kl.py:120-132, 263-275, 301-306) runs one sample at a time; every output row of the
    implicit-GEMM convolutions, every GroupNorm group and every attention row depends on its own sample only, so the result must
    equal the batched one bit for bit -- through the VAE object and through the pipeline's enable_vae_slicing()."""
    from oracle import vae_oracle as vo
    from textflux_amd.vae import AutoencoderKL
    kw = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)
    sd = vo.seeded_state_dict(vo.VaeConfig(**kw), 77)
    vae = AutoencoderKL(**kw).load_state_dict(sd, device="cuda")
    x = rnd((3, 3, 48, 40), 13).clamp(-1, 1).to(BF).cuda()
    z = rnd((3, 16, 12, 10), 14).to(BF).cuda()
    enc, dec = vae.encode(x).latent_dist.mean, vae.decode(z, return_dict=False)[0]
    vae.enable_slicing()
    try:
        enc_s, dec_s = vae.encode(x).latent_dist.mean, vae.decode(z, return_dict=False)[0]
    finally:
        vae.disable_slicing()
    assert torch.equal(enc, enc_s) and torch.equal(dec, dec_s)
    assert not torch.equal(dec[0], dec[1])                           # three different samples went through
    from textflux_amd.pipeline import FluxFillPipeline
    pipe = FluxFillPipeline.__new__(FluxFillPipeline)
    pipe.vae = vae
    pipe.enable_vae_slicing()
    assert vae.use_slicing
    pipe.disable_vae_slicing()
    assert not vae.use_slicing
    pipe.enable_vae_tiling()
    assert vae.use_tiling
    pipe.disable_vae_tiling()
    assert not vae.use_tiling


G9_VAE = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)


def test_vae_tiling_matches_the_references_tiled_encode_decode(golden):
    """enable_tiling() (reference: autoencoder_kl.py:145-160, 264-267, 301-303, 346-395, 456-503) against fixtures generated by the
    IMPORTED reference with tiling on (tests/golden/g13_vae_tiled.safetensors: sample_size 32 -> tiles of 32 px / 4 latent px every
    24 / 3, seams blended over 1 latent px / 8 px; 80 x 56 image = a ragged 4 x 3 tile grid with 8-px edge tiles, 10 x 7 latents): the
    HIP VAE runs every tile through the whole encoder / decoder, blends in the reference's order and rounding, crops and
    concatenates.  Compared with the reference's bf16 run at the bf16 level and with its fp32 run at the level the untiled path is;
    and the seam blend itself (tfx_blend_edge_nhwc) bit for bit against the reference's two bf16 tensor ops."""
    from oracle import vae_oracle as vo
    from textflux_amd import ops as o
    from textflux_amd.vae import AutoencoderKL
    g = golden("g13_vae_tiled")
    cfg = vo.VaeConfig(**G9_VAE)
    sd = vo.seeded_state_dict(cfg, 1300)
    vae = AutoencoderKL(sample_size=32, **G9_VAE).load_state_dict(sd, device="cuda")
    assert (vae.tile_sample_min_size, vae.tile_latent_min_size, vae.tile_overlap_factor) == (32, 4, 0.25)
    x, z = g["x"].to(BF).cuda(), g["z"].to(BF).cuda()
    untiled = vae.decode(z, return_dict=False)[0]
    close(untiled, g["f32.dec.untiled"], max_rel=5e-2, mae_rel=1.5e-2)
    vae.enable_tiling()
    try:
        post = vae.encode(x).latent_dist
        dec = vae.decode(z, return_dict=False)[0]
    finally:
        vae.disable_tiling()
    assert dec.shape == (2, 3, 80, 56) and post.mean.shape == (2, 16, 10, 7)
    close(post.mean, g["f32.enc.mean"], max_rel=5e-2, mae_rel=1.5e-2)
    close(post.std, g["f32.enc.std"], max_rel=5e-2, mae_rel=1.5e-2)
    close(dec, g["f32.dec.out"], max_rel=5e-2, mae_rel=1.5e-2)
    # two bf16 runs of one fp32 function: as close to each other as either is to fp32
    close(dec, g["bf16.dec.out"], max_rel=5e-2, mae_rel=1.5e-2)
    assert not torch.equal(dec, untiled)
    d_tiling = (g["f32.dec.out"] - g["f32.dec.untiled"]).abs().mean().item()
    assert (dec.float().cpu() - g["f32.dec.out"]).abs().mean().item() < 0.5 * d_tiling      # the tiled image, not the untiled one
    # the seam blend alone, bit-exact: a, b bf16 NHWC tiles, vertical and horizontal, extent clamped by a short tile
    a4, b4 = rnd((2, 9, 12, 16), 70).to(BF), rnd((2, 5, 12, 16), 71).to(BF)
    for axis, ext in ((1, 8), (1, 3), (2, 8), (2, 1)):
        aa, bb = (a4, b4) if axis == 1 else (a4.transpose(1, 2).contiguous(), b4.transpose(1, 2).contiguous())
        ref = bb.permute(0, 3, 1, 2).clone()
        (vo._blend_v if axis == 1 else vo._blend_h)(aa.permute(0, 3, 1, 2), ref, ext)
        got = o.blend_edge_nhwc_(aa.cuda(), bb.cuda().clone(), ext, axis)
        assert torch.equal(got.cpu().permute(0, 3, 1, 2), ref), (axis, ext)


def test_blend_edge_kernel_matches_the_references_own_blend_outputs(golden):
    """tfx_blend_edge_nhwc against AutoencoderKL.blend_v / blend_h outputs of the IMPORTED reference (fixture g15, bf16, the
    extent clamped by a short tile), bit for bit -- not through the oracle's restatement of the blend."""
    from textflux_amd import ops as o
    g = golden("g15_vae_blend")
    for ax, axis, exts in (("v", 1, (8, 3)), ("h", 2, (8, 1))):
        a, b = g[f"{ax}.a"].permute(0, 2, 3, 1).contiguous(), g[f"{ax}.b"].permute(0, 2, 3, 1).contiguous()      # NCHW -> NHWC
        for ext in exts:
            got = o.blend_edge_nhwc_(a.cuda(), b.cuda().clone(), ext, axis)
            assert torch.equal(got.cpu().permute(0, 3, 1, 2), g[f"{ax}.out.{ext}"]), (ax, ext)


def test_vae_rejects_configs_the_kernels_do_not_cover():
    from textflux_amd.vae import AutoencoderKL
    with pytest.raises(ValueError):
        AutoencoderKL(block_out_channels=(8, 16, 16, 16), layers_per_block=1, norm_num_groups=4)


def test_mid_block_attention_matches_fp32_softmax_attention():
    """The VAE mid-block attention (one head of dim C) as GEMM -> row softmax -> GEMM, against fp32 attention on the same
    bf16 q / k / v, at a token count that is not a tile multiple."""
    from textflux_amd import ops as o
    N, C = 35 * 29, 128
    q, k, v = (rnd((N, C), 30 + i, 1.5).to(BF) for i in range(3))
    ref = torch.softmax(q.float() @ k.float().T * C ** -0.5, -1) @ v.float()
    s = o.gemm_f32(q.cuda(), k.cuda())
    assert s.dtype == torch.float32
    close(s, q.float() @ k.float().T, max_rel=1e-5, mae_rel=1e-5)     # fp32 accumulators, unrounded
    Np = (N + 63) // 64 * 64
    pw = torch.zeros(N, Np, dtype=BF, device="cuda")
    o.row_softmax(s, C ** -0.5, pw)
    vt = torch.zeros(C, Np, dtype=BF, device="cuda")
    o.transpose(v.cuda()[None], out=vt[None, :, :N])
    out = o.gemm(pw, vt, None)
    close(out, ref.to(BF), max_rel=1.5e-2, mae_rel=3e-3)


def test_mid_block_attention_score_chunks_bound_the_scratch():
    """Round 5: the mid-block attention walks the query rows in chunks through one reused score / weight
    buffer pair whose size is capped (AutoencoderKL.ATTEND_CHUNK_BYTES) instead of materialising [N, N]: ragged token count, ragged last
    chunk, batch 2.  The scratch holds `rows` rows per image, not N; reruns are bit-identical; against one chunk per image the result differs only
    by the fp32 summation order of the K-sliced P v product (isolated last-ulp flips); both match fp32 softmax attention."""
    from textflux_amd.vae import AutoencoderKL
    vae = AutoencoderKL.__new__(AutoencoderKL)
    vae._scores = None
    B, N, C = 2, 35 * 61, 512
    q, k, v = ((rnd((B, N, C), 40 + i, 1.2).to(BF)).cuda() for i in range(3))
    vae.ATTEND_CHUNK_BYTES = 1 << 40
    whole = vae._attend(q, k, v)
    assert vae._scores[0].shape[:2] == (B, N)              # round 6: the whole batch per launch (per-batch weights: tfx_gemm_args.w_bstride)
    vae.ATTEND_CHUNK_BYTES = 8 << 20                       # 8 MiB / (2 x 2176 x 6 B) -> 256 rows per chunk: 9 chunks, the last one ragged
    parts = vae._attend(q, k, v)
    assert vae._scores[0].shape[:2] == (B, 256) and vae._scores[0].numel() * 6 <= 8 << 20
    assert torch.equal(parts, vae._attend(q, k, v))
    d = (parts.float() - whole.float()).abs()
    assert d.max().item() <= 2 ** -7 * whole.float().abs().max().item() and (d > 0).float().mean().item() < 0.05
    for got in (parts, whole):
        ref = torch.softmax(q[1].float() @ k[1].float().T * C ** -0.5, -1) @ v[1].float()
        close(got[1], ref.to(BF), max_rel=1.5e-2, mae_rel=3e-3)


@pytest.mark.parametrize("M,N,K,batch", [(300, 520, 128, 1), (1015, 1015, 256, 2), (256, 264, 64, 1)])
def test_gemm_fp32_output(M, N, K, batch):
    """Raw fp32 accumulators out of the persistent MFMA kernel (K % 128 == 0) and the generic kernel (K = 64), ragged
    M / N, row-strided output."""
    from textflux_amd import ops as o
    a, w = rnd((batch, M, K), 40).to(BF), rnd((N, K), 41).to(BF)
    ld = (N + 7) // 8 * 8 + 8
    buf = torch.full((batch, M, ld), 7.0, dtype=torch.float32, device="cuda")
    o.gemm_f32(a.cuda(), w.cuda(), out=buf[:, :, :N])
    ref = a.float() @ w.float().T
    assert (buf[:, :, :N].cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert torch.all(buf[:, :, N:] == 7.0)
