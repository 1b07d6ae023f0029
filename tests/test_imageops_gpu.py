"""GPU parity of the pixel / latent layout kernels around the VAE (imageops.hip) against the CPU restatement of the
reference's tensor ops (oracle/pipeline_oracle.py, oracle/vae_oracle.py) and the g6 layout goldens.  Integer / layout work
is bit-exact; the bf16 op chains are reproduced rounding for rounding (one bf16 ulp allowed where exp() is involved)."""
import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import vae_oracle as vo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from textflux_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def dev_scalar(t: torch.Tensor, op: str, c: float) -> torch.Tensor:
    """bf16 tensor (op) Python scalar as the reference's DEVICE kernels evaluate it: the scalar stays fp32 (opmath), a
    division becomes a multiply by the fp32 reciprocal, the result is rounded to bf16 once.  (torch's CPU kernels round
    the scalar to bf16 first; the kernels follow the GPU semantics -- see imageops.hip.)"""
    x, cf = t.float(), torch.tensor(c, dtype=torch.float32)
    r = {"sub": lambda: x - cf, "add": lambda: x + cf, "mul": lambda: x * cf,
         "div": lambda: x * (torch.tensor(1.0) / cf)}[op]()
    return r.to(BF)


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 representation steps (sign-magnitude -> ordered integers)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7fff), i)
    return (key(a.to(BF).cpu()) - key(b.to(BF).cpu())).abs()


@pytest.mark.parametrize("mask_batch", [1, 2])
@pytest.mark.parametrize("already_normalised", [False, True])
def test_prep_image_float_matches_processor_arithmetic(ops, mask_batch, already_normalised):
    B, H, W = 2, 32, 48
    img = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1))
    if already_normalised:
        img = img * 2 - 1          # has negatives: VaeImageProcessor.preprocess then skips the 2x-1 (IP:700-707)
    mask = torch.rand(mask_batch, 1, H, W, generator=torch.Generator().manual_seed(2))
    ref_img = img if already_normalised else 2.0 * img - 1.0
    m = mask.clone()
    m[m < 0.5] = 0
    m[m >= 0.5] = 1
    ref = (ref_img * (1 - m)).to(BF)                                   # P:2030-2031
    flag = ops.any_negative(img.cuda())
    got = ops.prep_image(img.cuda(), mask.cuda(), norm_mode=2, neg_flag=flag)
    assert got.shape == (B, H, W, 8) and int(flag.item()) == int(already_normalised)
    assert torch.equal(got[..., :3].cpu(), ref.permute(0, 2, 3, 1))
    assert torch.count_nonzero(got[..., 3:]).item() == 0


def test_prep_image_uint8_matches_pil_branch(ops):
    from textflux_amd.image_processor import VaeImageProcessor
    import PIL.Image
    rng = np.random.default_rng(3)
    imgs = [PIL.Image.fromarray(rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)) for _ in range(2)]
    masks = [PIL.Image.fromarray(rng.integers(0, 256, (32, 48), dtype=np.uint8)) for _ in range(2)]
    ip = VaeImageProcessor(vae_scale_factor=16)
    mp = VaeImageProcessor(vae_scale_factor=16, vae_latent_channels=16, do_normalize=False, do_binarize=True, do_convert_grayscale=True)
    ref = (ip.preprocess(imgs, 32, 48) * (1 - mp.preprocess(masks, 32, 48))).to(BF)       # the all-host arithmetic
    raw, rawm = ip.to_raw(imgs, 32, 48), mp.to_raw(masks, 32, 48)
    assert raw.dtype == torch.uint8 and raw.shape == (2, 32, 48, 3) and rawm.shape == (2, 32, 48)
    got = ops.prep_image(raw.cuda(), rawm.cuda(), norm_mode=1)
    assert torch.equal(got[..., :3].cpu(), ref.permute(0, 2, 3, 1))
    packed = torch.zeros(2, 6, 320, dtype=BF, device="cuda")
    ops.pack_mask(rawm.cuda(), packed, 64, 2, 32, 48, binarize=True)
    assert torch.equal(packed[..., 64:].cpu(), po.pack_mask(mp.preprocess(masks, 32, 48)).to(BF))
    assert torch.count_nonzero(packed[..., :64]).item() == 0


def test_pack_mask_matches_reference_golden(ops, golden):
    g = golden("g6_layout")
    m = g["mask.in"]                                                   # [2, 1, 64, 96] {0, 1}
    out = torch.empty(2, 24, 256, dtype=BF, device="cuda")
    ops.pack_mask(m.cuda(), out, 0, 2, 64, 96, binarize=False)
    assert torch.equal(out.float().cpu(), g["mask.out"].float())


def test_sample_pack_matches_bf16_reference_chain(ops):
    B, h, w, L = 2, 8, 12, 16
    mom = torch.cat([rnd((B, L, h, w), 4), rnd((B, L, h, w), 5, 2.0) - 1.0], 1).to(BF)      # NCHW moments (mean | logvar)
    eps = rnd((B, L, h, w), 6).to(BF)
    mean, logvar = mom.chunk(2, 1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))            # bf16 tensor ops = the reference's rounding points
    z = mean + std * eps
    ref = po.pack_latents(dev_scalar(dev_scalar(z, "sub", 0.1159), "mul", 0.3611))
    out = torch.zeros(B, (h // 2) * (w // 2), 320, dtype=BF, device="cuda")
    ops.vae_sample_pack(mom.permute(0, 2, 3, 1).contiguous().cuda(), eps.cuda(), out, 0, 0.1159, 0.3611)
    d = ulp_diff(out[..., :64], ref)
    assert d.max().item() <= 1 and (d > 0).float().mean().item() < 0.02, (d.max().item(), (d > 0).float().mean().item())
    assert torch.count_nonzero(out[..., 64:]).item() == 0
    # mode (no eps): exact
    out2 = torch.zeros(B, (h // 2) * (w // 2), 64, dtype=BF, device="cuda")
    ops.vae_sample_pack(mom.permute(0, 2, 3, 1).contiguous().cuda(), None, out2, 0, 0.1159, 0.3611)
    assert torch.equal(out2.cpu(), po.pack_latents(dev_scalar(dev_scalar(mean, "sub", 0.1159), "mul", 0.3611)))


def test_unpack_latents_matches_reference_chain(ops, golden):
    g = golden("g6_layout")
    lat = rnd((2, 24, 64), 7).to(BF)
    ref = dev_scalar(dev_scalar(po.unpack_latents(lat, 64, 96), "div", 0.3611), "add", 0.1159)      # P:2126-2127
    got = ops.unpack_latents(lat.cuda(), 8, 12, 0.1159, 0.3611)
    assert got.shape == (2, 8, 12, 16)
    assert torch.equal(got.cpu(), ref.permute(0, 2, 3, 1))
    # the layout itself against the reference's own _pack_latents -> _unpack_latents golden (shift 0, scale 1: pure layout)
    lay = ops.unpack_latents(g["pack.out"].to(BF).cuda(), 8, 12, 0.0, 1.0)
    assert torch.equal(lay.cpu(), g["unpack.out"].to(BF).permute(0, 2, 3, 1))


@pytest.mark.parametrize("denorm", [True, False])
def test_postprocess_modes(ops, denorm):
    B, H, W = 2, 16, 24
    x = rnd((B, H, W, 8), 8, 0.8).to(BF)
    img = x[..., :3].permute(0, 3, 1, 2)                               # the NCHW bf16 tensor the reference's decode returns
    den = (img / 2 + 0.5).clamp(0, 1) if denorm else img               # bf16 ops
    pt = ops.postprocess(x.cuda(), 3, "pt", denorm)
    assert torch.equal(pt.cpu(), den)
    npv = ops.postprocess(x.cuda(), 3, "np", denorm)
    assert torch.equal(npv.cpu(), den.permute(0, 2, 3, 1).float())
    if denorm:
        u8 = ops.postprocess(x.cuda(), 3, "u8", denorm)
        ref = (den.permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")
        assert np.array_equal(u8.cpu().numpy(), ref)
        box = (5, 3, 21, 14)                                           # PIL box (left, top, right, bottom)
        u8c = ops.postprocess(x.cuda(), 3, "u8", denorm, crop=box)
        assert np.array_equal(u8c.cpu().numpy(), ref[:, 3:14, 5:21])
        ptc = ops.postprocess(x.cuda(), 3, "pt", denorm, crop=box)
        assert torch.equal(ptc.cpu(), den[:, :, 3:14, 5:21])


def test_transpose_and_row_softmax(ops):
    x = rnd((3, 100, 72), 9).to(BF)
    assert torch.equal(ops.transpose(x.cuda()).cpu(), x.transpose(1, 2))
    for N in (96, 1001, 4096):
        s = rnd((37, N), 10, 30.0)
        ref = torch.softmax(s * 0.0442, dim=-1)
        out = torch.zeros(37, N + 8, dtype=BF, device="cuda")
        got = ops.row_softmax(s.cuda(), 0.0442, out).float().cpu()
        assert (got[:, :N] - ref).abs().max().item() <= 2 ** -8 * ref.max().item() + 1e-7
        assert torch.count_nonzero(got[:, N:]).item() == 0 and abs(got.sum(-1) - 1).max().item() < 5e-3


def test_compose_canvas_is_the_host_stacking_bit_for_bit():
    """tfx_compose_canvas against the host composition of the reference's callers: numpy stacking of glyph + scene, a black
    mask over the glyph part, PIL's convert("L") of the RGB mask (random grey-ish RGB values, not only black / white)."""
    import numpy as np
    from PIL import Image
    from textflux_amd import ops
    rng = np.random.default_rng(3)
    for horizontal, (gh, gw, sh, sw) in ((False, (40, 96, 64, 96)), (True, (64, 48, 64, 80))):
        B = 3
        g = rng.integers(0, 256, (B, gh, gw, 3), dtype=np.uint8)
        s = rng.integers(0, 256, (B, sh, sw, 3), dtype=np.uint8)
        m = rng.integers(0, 256, (B, sh, sw, 3), dtype=np.uint8)
        canvas, cmask = ops.compose_canvas(torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(m).cuda(), horizontal)
        stack = np.hstack if horizontal else np.vstack
        for b in range(B):
            ref_img = stack((g[b], s[b]))
            ref_mask = np.array(Image.fromarray(stack((np.zeros_like(g[b]), m[b]))).convert("L"))
            assert np.array_equal(canvas[b].cpu().numpy(), ref_img)
            assert np.array_equal(cmask[b].cpu().numpy(), ref_mask)


def test_pipeline_conditioning_from_a_device_canvas_equals_the_pil_path():
    """FluxFillPipeline._encode_conditioning fed with the device-composed uint8 canvas gives bit-identical conditioning
    latents to the PIL inputs the reference's callers build on the host."""
    import numpy as np
    from PIL import Image
    from textflux_amd import ops
    from textflux_amd.pipeline import FluxFillPipeline
    from textflux_amd.vae import AutoencoderKL
    from textflux_amd.schedulers import FlowMatchEulerDiscreteScheduler
    vae = AutoencoderKL(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16).init_random_(seed=2, device="cuda")
    pipe = FluxFillPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), vae=vae, text_encoder=None, tokenizer=None, text_encoder_2=None,
                            tokenizer_2=None, transformer=None)
    pipe._device = torch.device("cuda")
    rng = np.random.default_rng(4)
    B, gh, sh, W = 2, 32, 96, 64
    g = rng.integers(0, 256, (B, gh, W, 3), dtype=np.uint8)
    s = rng.integers(0, 256, (B, sh, W, 3), dtype=np.uint8)
    m = np.zeros((B, sh, W, 3), np.uint8)
    m[:, 20:60, 8:40] = 255
    m[:, 70:80, 10:50] = rng.integers(0, 256, (B, 10, 40, 3), dtype=np.uint8)        # grey values around the binarisation threshold
    pil_i = [Image.fromarray(np.vstack((g[b], s[b]))) for b in range(B)]
    pil_m = [Image.fromarray(np.vstack((np.zeros_like(g[b]), m[b]))) for b in range(B)]
    canvas, cmask = ops.compose_canvas(torch.from_numpy(g).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(m).cuda(), False)
    H = gh + sh
    outs = []
    for img, msk in ((pil_i, pil_m), (canvas, cmask)):
        gen = torch.Generator(device="cuda").manual_seed(7)
        cond, h, w = pipe._encode_conditioning(img, msk, H, W, B, 1, torch.bfloat16, "cuda", gen)
        assert (h, w) == (H, W)
        outs.append(cond)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("h,w,ho,wo", [(168, 256, 160, 256), (592, 512, 576, 512), (200, 330, 192, 320), (100, 90, 128, 96), (77, 131, 64, 160)])
def test_resample_u8_is_pillow_bicubic_bit_for_bit(h, w, ho, wo):
    """tfx_resample_u8 with the host-built coefficient tables against PIL.Image.resize (default filter), RGB and grey, down-
    and up-scaling on either axis."""
    import numpy as np
    from PIL import Image
    from textflux_amd import ops
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    got = ops.resample_u8(torch.from_numpy(a).cuda(), (ho, wo)).cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], np.array(Image.fromarray(a[b]).resize((wo, ho))))
    grey = ops.rgb_to_grey(torch.from_numpy(a).cuda()).cpu().numpy()
    for b in range(2):
        assert np.array_equal(grey[b], np.array(Image.fromarray(a[b]).convert("L")))
