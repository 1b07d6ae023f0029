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


def test_vae_hip_nhwc_path_matches_oracle():
    """Whole encoder / decoder on the HIP NHWC path (widths 64 / 128 so every conv takes the MFMA kernel) against the
    fp32 CPU oracle, and against the torch/MIOpen NCHW path of the same class."""
    from oracle import vae_oracle as vo
    from textflux_amd.vae import AutoencoderKL
    kw = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)
    cfg = vo.VaeConfig(**kw)
    sd = vo.seeded_state_dict(cfg, 321)
    vae = AutoencoderKL(**kw).load_state_dict(sd, device="cuda")
    assert vae.use_hip
    x = rnd((2, 3, 48, 40), 11).clamp(-1, 1)
    z = rnd((2, 16, 12, 10), 12)
    mean, std = vo.encode_moments(x, sd, cfg)
    post = vae.encode(x.to(BF).cuda()).latent_dist
    dec_ref = vo.decoder(z, sd, cfg)
    dec = vae.decode(z.to(BF).cuda(), return_dict=False)[0]
    assert dec.shape == dec_ref.shape
    close(post.mean, mean, max_rel=5e-2, mae_rel=1.5e-2)
    close(dec, dec_ref, max_rel=5e-2, mae_rel=1.5e-2)
    vae.use_hip = False
    dec_t = vae.decode(z.to(BF).cuda(), return_dict=False)[0]
    e_hip = (dec.float().cpu() - dec_ref).abs().mean().item()
    e_torch = (dec_t.float().cpu() - dec_ref).abs().mean().item()
    print(f"decoder MAE vs fp32 oracle: HIP path {e_hip:.3e}, torch/MIOpen path {e_torch:.3e}")
    assert e_hip < 2.0 * e_torch + 1e-3
