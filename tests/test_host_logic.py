"""CPU tests of the host-side logic of the product package (no GPU, no kernels): scheduler tables vs the reference
goldens, coefficient tables, pipeline helpers, input validation, image processor, driver geometry / prompts."""
import numpy as np
import pytest
import torch
from PIL import Image

from textflux_amd import glyph
from textflux_amd.image_processor import VaeImageProcessor
from textflux_amd.pipeline import FluxFillPipeline, calculate_shift, randn_tensor, retrieve_timesteps
from textflux_amd.schedulers import FlowMatchEulerDiscreteScheduler, StochasticRFOvershotDiscreteScheduler
from textflux_amd.transformer import FluxTransformer2DModel, rope_tables

SCHED = dict(use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096,
             shift=3.0)


@pytest.mark.parametrize("n", [4, 30, 50])
@pytest.mark.parametrize("S", [1152, 4096, 4736, 8192])
def test_scheduler_tables_match_reference(golden, n, S):
    g = golden("g4_sched")
    mu = calculate_shift(S, 256, 4096, 0.5, 1.15)
    assert mu == g[f"mu.S{S}"].item()
    e = FlowMatchEulerDiscreteScheduler(**SCHED)
    ts, k = retrieve_timesteps(e, n, "cpu", sigmas=np.linspace(1.0, 1 / n, n), mu=mu)
    assert k == n and torch.equal(e.sigmas, g[f"euler.n{n}.S{S}.sigmas"]) and torch.equal(ts, g[f"euler.n{n}.S{S}.timesteps"])
    a = StochasticRFOvershotDiscreteScheduler.from_config(e.config)  # config hand-over as run_inference.py:81-82
    a.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu)
    assert torch.equal(a.sigmas, g[f"amo.n{n}.S{S}.sigmas"]) and torch.equal(a.timesteps, g[f"amo.n{n}.S{S}.timesteps"])
    assert e.order == 1 and a.order == 1 and len(e) == 1000


def test_coefficient_tables():
    n, S = 30, 4096
    mu = calculate_shift(S, 256, 4096, 0.5, 1.15)
    e = FlowMatchEulerDiscreteScheduler(**SCHED)
    e.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu)
    c = e.coef_table("cpu")
    assert c.shape == (n,) and (c < 0).all()
    assert torch.equal(c, c.to(torch.bfloat16).float())           # bf16-rounded, as the reference's multiply sees it
    assert abs(c.sum().item() + 1.0) < 2e-2                       # sigma goes 1 -> 0
    a = StochasticRFOvershotDiscreteScheduler(**SCHED)
    with pytest.raises(RuntimeError):
        a.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu) or a.coef_table("cpu")
    a.set_c(2.0)
    a.set_overshot_func(lambda t, dt: t + dt)
    t = a.coef_table("cpu")
    assert t.shape == (n, 3) and (t[:, 1] <= 1).all() and (t[:, 2] >= 0).all()
    assert t[-1, 1].item() == 1.0 and t[-1, 2].item() == 0.0      # last step deterministic
    with pytest.raises(ValueError):
        FlowMatchEulerDiscreteScheduler(**SCHED).set_timesteps(num_inference_steps=4)   # dynamic shifting needs mu


def test_layout_helpers_match_reference(golden):
    g = golden("g6_layout")
    P = FluxFillPipeline
    assert torch.equal(P._pack_latents(g["pack.in"], 2, 16, 8, 12), g["pack.out"])
    assert torch.equal(P._unpack_latents(g["pack.out"], 64, 96, 8), g["unpack.out"])
    assert torch.equal(P._prepare_latent_image_ids(1, 4, 6, "cpu", torch.float32), g["ids.4x6"])
    g1 = golden("g1_ops")
    cos, sin = rope_tables(g1["rope.ids"])
    assert torch.equal(cos, g1["rope.cos"]) and torch.equal(sin, g1["rope.sin"])


class _FakeVae:
    class config:
        block_out_channels = (1, 1, 1, 1)
        latent_channels = 16
        scaling_factor = 0.3611
        shift_factor = 0.1159


def _pipe():
    tr = FluxTransformer2DModel(in_channels=384, out_channels=64, num_layers=1, num_single_layers=1,
                                num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32, guidance_embeds=True)
    return FluxFillPipeline(scheduler=FlowMatchEulerDiscreteScheduler(**SCHED), vae=_FakeVae(), text_encoder=None,
                            tokenizer=None, text_encoder_2=None, tokenizer_2=None, transformer=tr)


def test_check_inputs_raises_like_the_reference():
    p = _pipe()
    pe, pooled = torch.zeros(1, 4, 64), torch.zeros(1, 32)
    ok = dict(prompt=None, prompt_2=None, height=64, width=64, prompt_embeds=pe, pooled_prompt_embeds=pooled)
    p.check_inputs(**ok)
    for bad in (dict(prompt="x"), dict(prompt_2="x"), dict(prompt_embeds=None), dict(pooled_prompt_embeds=None),
                dict(max_sequence_length=513), dict(callback_on_step_end_tensor_inputs=["nope"]),
                dict(image=torch.zeros(1, 3, 8, 8)), dict(image=torch.zeros(1, 3, 8, 8), mask_image=1, masked_image_latents=1)):
        with pytest.raises(ValueError):
            p.check_inputs(**{**ok, **bad})
    assert p.vae_scale_factor == 8 and p.tokenizer_max_length == 77 and p.default_sample_size == 128


def test_randn_tensor_generator_semantics():
    a = randn_tensor((2, 3), generator=torch.Generator().manual_seed(1))
    b = randn_tensor((2, 3), generator=torch.Generator().manual_seed(1))
    assert torch.equal(a, b)
    gens = [torch.Generator().manual_seed(5), torch.Generator().manual_seed(6)]
    c = randn_tensor((2, 3), generator=gens)
    assert torch.equal(c[0:1], randn_tensor((1, 3), generator=torch.Generator().manual_seed(5)))


def test_image_processor():
    ip = VaeImageProcessor(vae_scale_factor=16)
    img = Image.fromarray((np.arange(40 * 50 * 3) % 256).astype(np.uint8).reshape(40, 50, 3))
    t = ip.preprocess(img, height=40, width=50)
    assert t.shape == (1, 3, 32, 48) and t.min() >= -1 and t.max() <= 1
    mp = VaeImageProcessor(vae_scale_factor=16, vae_latent_channels=16, do_normalize=False, do_binarize=True,
                           do_convert_grayscale=True)
    m = mp.preprocess(Image.fromarray(np.full((32, 32, 3), 200, np.uint8)), height=32, width=32)
    assert m.shape == (1, 1, 32, 32) and set(m.unique().tolist()) == {1.0}
    x = torch.linspace(-1.2, 1.2, 3 * 4 * 4).reshape(1, 3, 4, 4)
    out = ip.postprocess(x, "np")
    assert out.shape == (1, 4, 4, 3) and out.min() == 0 and out.max() == 1
    assert ip.postprocess(x, "pil")[0].size == (4, 4) and ip.postprocess(x, "latent") is x


@pytest.mark.parametrize("w,h,multi,pipe,crop", [
    (512, 512, False, (512, 576), (0, 77, 512, 576)), (512, 512, True, (512, 1024), (0, 512, 512, 1024)),
    (1024, 1024, False, (1024, 1184), (0, 160, 1024, 1184)), (1024, 512, False, (1024, 672), (0, 160, 1024, 672)),
    (1000, 700, False, (992, 832), (0, 151, 992, 832)), (700, 1000, True, (1376, 992), (688, 0, 1376, 992)),
    (1400, 900, False, (1376, 1088), (0, 212, 1376, 1088))])
def test_driver_geometry_known_answers(w, h, multi, pipe, crop):
    scene, mask, words = glyph.synthetic_case(w, h, multiline=multi)
    combined, cmask, meta = glyph.compose(scene, mask, words)
    assert combined.size == cmask.size
    assert glyph.pipe_size(combined) == pipe
    assert glyph.crop_box(pipe, meta) == crop
    arr = np.array(cmask)
    if meta["direction"] == "vertical":
        assert arr[: arr.shape[0] - h].max() == 0     # the glyph half of the mask is black
    else:
        assert arr[:, : arr.shape[1] - w].max() == 0


def test_prompts_verbatim():
    assert glyph.generate_prompt(["陕西"]).endswith("[IMAGE2] shows the text content '陕西' naturally and correspondingly integrated into the image.")
    assert "with the words 'a', 'b';" in glyph.generate_prompt(["a", "b"])
    assert glyph.PROMPT_TEMPLATE2.startswith("The pair of images highlights some white words on a black background")
    assert glyph.read_words_from_text(" x \n\n y ") == ["x", "y"]


def test_state_dict_key_contract():
    from oracle import flux_oracle as fo
    cfg = fo.FluxConfig(num_layers=2, num_single_layers=3, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    m = FluxTransformer2DModel(in_channels=384, out_channels=64, num_layers=2, num_single_layers=3, num_attention_heads=2,
                               joint_attention_dim=64, pooled_projection_dim=32, guidance_embeds=True)
    assert sorted(m.expected_keys()) == sorted(fo.state_dict_shapes(cfg).keys())   # = the reference's key set (G-goldens)
    assert m.mod_len == 12 * 256 * 2 + 3 * 256 * 3 + 2 * 256
    full = FluxTransformer2DModel(in_channels=384, out_channels=64, guidance_embeds=True)
    assert full.mod_len == 1056768 and len(full.expected_keys()) == 1160
    with pytest.raises(ValueError):
        FluxTransformer2DModel(attention_head_dim=64)


def test_encode_prompt_with_tiny_text_encoders():
    """encode_prompt / _get_clip_prompt_embeds / _get_t5_prompt_embeds (pipeline_flux_fill.py:1411-1503, 1586-1663) against
    tiny random CLIP / T5 encoders from `transformers` configs (no hub) and stub tokenizers: CLIP pooled output of
    `prompt`, T5 sequence of `prompt_2`, repeat semantics of num_images_per_prompt, zero text ids."""
    from types import SimpleNamespace
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

    class Tok:
        def __init__(self, n):
            self.model_max_length = n

        def __call__(self, prompts, padding=None, max_length=None, truncation=None, return_tensors=None, **kw):
            ids = torch.stack([torch.tensor([(hash(p) + i) % 90 + 1 for i in range(max_length)]) for p in prompts])
            return SimpleNamespace(input_ids=ids)

    torch.manual_seed(0)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                        num_attention_heads=2, max_position_embeddings=77, projection_dim=32)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=100, d_model=64, d_kv=16, d_ff=64, num_layers=1, num_heads=2)).eval()
    p = _pipe()
    p.text_encoder, p.tokenizer, p.text_encoder_2, p.tokenizer_2 = clip, Tok(77), t5, Tok(512)
    p.tokenizer_max_length = 77
    pe, pooled, ids = p.encode_prompt(prompt=["a", "b"], prompt_2=["c d", "e"], device="cpu", num_images_per_prompt=2,
                                      max_sequence_length=24)
    assert pe.shape == (4, 24, 64) and pooled.shape == (4, 32) and ids.shape == (24, 3) and ids.abs().sum() == 0
    assert torch.equal(pe[0], pe[1]) and not torch.equal(pe[0], pe[2])          # repeated per prompt, distinct across prompts
    assert torch.equal(pooled[2], pooled[3]) and not torch.equal(pooled[0], pooled[2])
    ref_pooled = clip(p.tokenizer(["a"], max_length=77).input_ids, output_hidden_states=False).pooler_output
    assert torch.allclose(pooled[0], ref_pooled[0])
    pe2, pooled2, _ = p.encode_prompt(prompt=None, prompt_2=None, prompt_embeds=pe, pooled_prompt_embeds=pooled, device="cpu")
    assert pe2 is pe and pooled2 is pooled
    # prompt cache: same results, the encoders only see strings they have not seen before
    calls = {"clip": 0, "t5": 0}
    clip_fwd, t5_fwd = clip.forward, t5.forward
    clip.forward = lambda *a, **k: (calls.__setitem__("clip", calls["clip"] + a[0].shape[0]), clip_fwd(*a, **k))[1]
    t5.forward = lambda *a, **k: (calls.__setitem__("t5", calls["t5"] + a[0].shape[0]), t5_fwd(*a, **k))[1]
    p.enable_prompt_cache(8)
    pe3, pooled3, _ = p.encode_prompt(prompt=["a", "b"], prompt_2=["c d", "e"], device="cpu", num_images_per_prompt=2,
                                      max_sequence_length=24)
    assert torch.equal(pe3, pe) and torch.equal(pooled3, pooled) and calls == {"clip": 2, "t5": 2}
    pe4, pooled4, _ = p.encode_prompt(prompt=["a", "a", "b"], prompt_2=["e", "new", "c d"], device="cpu", max_sequence_length=24)
    assert calls == {"clip": 2, "t5": 3}                                        # only "new" reached an encoder
    assert torch.equal(pe4[0], pe[2]) and torch.equal(pe4[2], pe[0]) and torch.equal(pooled4[0], pooled4[1])
    p.enable_prompt_cache(0)
    p.encode_prompt(prompt=["a"], prompt_2=["e"], device="cpu", max_sequence_length=24)
    assert calls == {"clip": 3, "t5": 4}
    clip.forward, t5.forward = clip_fwd, t5_fwd
    p.text_encoder = None
    with pytest.raises(ValueError):
        p.encode_prompt(prompt="a", prompt_2=None, device="cpu")


def test_pil_resample_tables_reproduce_pillow_bicubic():
    """image_processor.pil_resample_tables + the two integer passes tfx_resample_u8 runs (restated in numpy) == PIL.Image.resize,
    bit for bit: the host half of the device-side resize of the batch driver."""
    import numpy as np
    from PIL import Image
    from textflux_amd.image_processor import pil_resample_tables

    def one_pass(a, bounds, kk, axis):
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        out = np.empty((len(bounds),) + a.shape[1:], np.int64)
        for xx, (xmin, xmax) in enumerate(bounds):
            ss = np.full(a.shape[1:], 1 << 21, np.int64)
            for x in range(xmax):
                ss += a[xmin + x] * int(kk[xx, x])
            out[xx] = np.clip(ss >> 22, 0, 255)
        return np.moveaxis(out, 0, axis).astype(np.uint8)

    rng = np.random.default_rng(1)
    for (h, w, ho, wo) in [(168, 256, 160, 256), (74, 100, 64, 96), (50, 60, 64, 64)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        x = a
        if wo != w:
            x = one_pass(x, *pil_resample_tables(w, wo), axis=1)
        if ho != h:
            x = one_pass(x, *pil_resample_tables(h, ho), axis=0)
        assert np.array_equal(x, np.array(Image.fromarray(a).resize((wo, ho))))


def test_fill_polygon_slanted_and_concave_properties():
    """glyph.fill_polygon (reference: cv2.fillPoly, scripts/run_eval.py:92-96) on slanted and concave polygons: the properties
    OpenCV's rasteriser and this one share -- vertices white, every pixel whose centre is strictly inside white, every pixel
    further than one pixel from the polygon black, all three channels equal."""
    import numpy as np
    from textflux_amd import glyph

    def inside(px, py, poly):        # even-odd rule, point strictly inside
        n, c = len(poly), False
        for i in range(n):
            (x0, y0), (x1, y1) = poly[i], poly[(i + 1) % n]
            if (y0 > py) != (y1 > py) and px < (x1 - x0) * (py - y0) / (y1 - y0) + x0:
                c = not c
        return c

    def dist_to_poly(px, py, poly):
        best = 1e9
        for i in range(len(poly)):
            (x0, y0), (x1, y1) = poly[i], poly[(i + 1) % len(poly)]
            dx, dy = x1 - x0, y1 - y0
            t = 0.0 if dx == dy == 0 else max(0.0, min(1.0, ((px - x0) * dx + (py - y0) * dy) / (dx * dx + dy * dy)))
            best = min(best, ((px - x0 - t * dx) ** 2 + (py - y0 - t * dy) ** 2) ** 0.5)
        return best

    polys = [[[10, 5], [90, 20], [70, 60], [15, 45]],                         # slanted convex quadrilateral
             [[5, 5], [95, 8], [50, 30], [92, 62], [8, 58], [30, 30]],         # concave (two notches)
             [[20.9, 10.2], [80.7, 15.9], [60.1, 55.5]]]                       # float vertices truncate like np.int32
    for poly in polys:
        m = glyph.fill_polygon(70, 100, poly)
        assert m.shape == (70, 100, 3) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 255}
        assert (m[..., 0] == m[..., 1]).all() and (m[..., 0] == m[..., 2]).all()
        ip = [(int(x), int(y)) for x, y in poly]
        for x, y in ip:
            assert m[y, x, 0] == 255
        for y in range(70):
            for x in range(100):
                if inside(x, y, ip) and dist_to_poly(x, y, ip) > 0.75:
                    assert m[y, x, 0] == 255, (x, y)
                if not inside(x, y, ip) and dist_to_poly(x, y, ip) > 1.0:
                    assert m[y, x, 0] == 0, (x, y)
