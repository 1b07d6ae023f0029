#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Inputs are seeded torch-CPU tensors, weights come from oracle.flux_oracle.seeded_state_dict
loaded (strict) into the reference modules, outputs are whatever the reference computes.  Only data
(inputs + expected outputs) is written; no reference source is copied.  Golden groups follow
SURVEY.md §8(c): G1 per-op, G2 blocks, G3 model, G4 schedulers, G5 pipeline, G6 layout, G7 rounding.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/diffusers/src")
sys.dont_write_bytecode = True

import numpy as np
import torch
import transformers.utils as tu

tu.FLAX_WEIGHTS_NAME = "flax_model.msgpack"  # removed in transformers 5.x; the reference imports it

from safetensors.torch import save_file

import diffusers  # the reference, 0.32.0.dev0
from diffusers import FlowMatchEulerDiscreteScheduler, StochasticRFOvershotDiscreteScheduler
from diffusers.models.attention import FeedForward
from diffusers.models.embeddings import (CombinedTimestepGuidanceTextProjEmbeddings, FluxPosEmbed,
                                         apply_rotary_emb, get_timestep_embedding)
from diffusers.models.normalization import RMSNorm
from diffusers.models.transformers.transformer_flux import (FluxSingleTransformerBlock, FluxTransformer2DModel,
                                                            FluxTransformerBlock)
from diffusers.pipelines.flux.pipeline_flux_fill import FluxFillPipeline, calculate_shift

from oracle import flux_oracle as fo

OUT = os.path.dirname(os.path.abspath(__file__))
assert diffusers.__version__ == "0.32.0.dev0"
torch.manual_seed(0)
torch.set_grad_enabled(False)


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def sub_sd(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def save(name, tensors):
    tensors = {k: v.contiguous() for k, v in tensors.items()}
    save_file(tensors, os.path.join(OUT, name + ".safetensors"))
    n = sum(v.numel() * v.element_size() for v in tensors.values())
    print(f"{name}: {len(tensors)} tensors, {n/1e6:.2f} MB")


def ref_model(cfg: fo.FluxConfig, seed, dtype=torch.float32):
    m = FluxTransformer2DModel(
        patch_size=cfg.patch_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
        num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
        attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
        joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim,
        guidance_embeds=cfg.guidance_embeds, axes_dims_rope=cfg.axes_dims_rope)
    sd = fo.seeded_state_dict(cfg, seed)
    assert list(m.state_dict().keys()) == list(sd.keys()), "oracle key order != reference key order"
    m.load_state_dict(sd, strict=True)
    return m.to(dtype).eval(), sd


# ----------------------------------------------------------------------------- G1 per-op
def g1_ops():
    out = {}
    t = torch.tensor([0.0, 1.0, 250.5, 892.0, 1000.0, 29952.0])
    out["tsemb.t"] = t
    out["tsemb.out"] = get_timestep_embedding(t, 256, flip_sin_to_cos=True, downscale_freq_shift=0)
    # rope
    ids = torch.cat([torch.zeros(5, 3), fo_ids(4, 6)], 0)
    cos, sin = FluxPosEmbed(theta=10000, axes_dim=[16, 56, 56])(ids)
    out["rope.ids"], out["rope.cos"], out["rope.sin"] = ids, cos, sin
    x = rnd((2, 3, 29, 128), 11)
    out["rope.x"] = x
    out["rope.out_f32"] = apply_rotary_emb(x, (cos, sin))
    out["rope.out_bf16"] = apply_rotary_emb(x.bfloat16(), (cos, sin))
    # rmsnorm
    w = 1 + 0.1 * rnd((128,), 12)
    n = RMSNorm(128, eps=1e-6)
    n.weight.data.copy_(w)
    out["rms.w"] = w
    out["rms.out_f32"] = n(x)
    out["rms.out_bf16"] = n.to(torch.bfloat16)(x.bfloat16())
    # feed-forward (gelu tanh), D=256
    ff = FeedForward(dim=256, dim_out=256, activation_fn="gelu-approximate")
    ffsd = {k: rnd(v.shape, 13 + i, 0.05) for i, (k, v) in enumerate(ff.state_dict().items())}
    ff.load_state_dict(ffsd)
    xf = rnd((2, 7, 256), 14)
    out["ff.x"] = xf
    for k, v in ffsd.items():
        out["ff.sd." + k] = v
    out["ff.out_f32"] = ff(xf)
    out["ff.out_bf16"] = ff.to(torch.bfloat16)(xf.bfloat16())
    # time-text embedding
    te = CombinedTimestepGuidanceTextProjEmbeddings(embedding_dim=256, pooled_projection_dim=32)
    tesd = {k: rnd(v.shape, 30 + i, 0.05) for i, (k, v) in enumerate(te.state_dict().items())}
    te.load_state_dict(tesd)
    pooled = rnd((3, 32), 15)
    tt = torch.tensor([892.0, 500.0, 33.5])
    gg = torch.tensor([29952.0, 3500.0, 1000.0])
    for k, v in tesd.items():
        out["tte.sd.time_text_embed." + k] = v
    out["tte.t"], out["tte.g"], out["tte.pooled"] = tt, gg, pooled
    out["tte.out_f32"] = te(tt, gg, pooled)
    out["tte.out_bf16"] = te.to(torch.bfloat16)(tt.bfloat16(), gg.bfloat16(), pooled.bfloat16())
    save("g1_ops", out)


def fo_ids(h2, w2):
    return FluxFillPipeline._prepare_latent_image_ids(1, h2, w2, "cpu", torch.float32)


# ----------------------------------------------------------------------------- G2 blocks
def g2_blocks():
    out = {}
    for tag, heads, S, T, seed in (("d256", 2, 24, 8, 100), ("d3072", 24, 64, 32, 101)):
        D = heads * 128
        cfg = fo.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=heads,
                            joint_attention_dim=64, pooled_projection_dim=32)
        sd = fo.seeded_state_dict(cfg, seed)
        dbl = FluxTransformerBlock(D, heads, 128)
        dbl.load_state_dict(sub_sd(sd, "transformer_blocks.0."))
        sgl = FluxSingleTransformerBlock(D, heads, 128)
        sgl.load_state_dict(sub_sd(sd, "single_transformer_blocks.0."))
        hidden, enc, temb = rnd((2, S, D), seed + 1), rnd((2, T, D), seed + 2), rnd((2, D), seed + 3)
        h2, w2 = (4, 6) if S == 24 else (8, 8)
        ids = torch.cat([torch.zeros(T, 3), fo_ids(h2, w2)], 0)
        rope = FluxPosEmbed(theta=10000, axes_dim=[16, 56, 56])(ids)
        e_o, h_o = dbl(hidden_states=hidden, encoder_hidden_states=enc, temb=temb, image_rotary_emb=rope)
        joint = torch.cat([enc, hidden], 1)
        s_o = sgl(hidden_states=joint, temb=temb, image_rotary_emb=rope)
        keep_in = tag == "d256"
        if keep_in:
            out[f"{tag}.hidden"], out[f"{tag}.enc"], out[f"{tag}.temb"] = hidden, enc, temb
        out[f"{tag}.meta"] = torch.tensor([heads, S, T, seed, h2, w2])
        out[f"{tag}.double.enc_out"], out[f"{tag}.double.hidden_out"] = e_o, h_o
        out[f"{tag}.single.out"] = s_o
        if tag == "d256":
            eb, hb = dbl.to(torch.bfloat16)(hidden_states=hidden.bfloat16(), encoder_hidden_states=enc.bfloat16(),
                                            temb=temb.bfloat16(), image_rotary_emb=rope)
            out[f"{tag}.double.enc_out_bf16"], out[f"{tag}.double.hidden_out_bf16"] = eb, hb
            out[f"{tag}.single.out_bf16"] = sgl.to(torch.bfloat16)(hidden_states=joint.bfloat16(),
                                                                    temb=temb.bfloat16(), image_rotary_emb=rope)
    save("g2_blocks", out)


# ----------------------------------------------------------------------------- G3 model
G3_CFG = fo.FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64,
                       pooled_projection_dim=32)


def g3_inputs(B=2, S=64, T=16, seed=200):
    return dict(
        hidden_states=rnd((B, S, 384), seed), encoder_hidden_states=rnd((B, T, 64), seed + 1),
        pooled_projections=rnd((B, 32), seed + 2), timestep=torch.tensor([0.892, 0.25])[:B],
        guidance=torch.tensor([30.0, 3.5])[:B], img_ids=fo_ids(8, 8), txt_ids=torch.zeros(T, 3))


def g3_model():
    out = {}
    m, _ = ref_model(G3_CFG, 7)
    inp = g3_inputs()
    for k, v in inp.items():
        out["in." + k] = v
    out["out_f32"] = m(**inp, return_dict=False)[0]
    mb = m.to(torch.bfloat16)
    inb = {k: (v.bfloat16() if v.dtype == torch.float32 else v) for k, v in inp.items()}
    inb["guidance"] = inp["guidance"]  # the pipeline passes guidance in f32 (pipeline_flux_fill.py:2070)
    out["out_bf16"] = mb(**inb, return_dict=False)[0]
    save("g3_model", out)


# ----------------------------------------------------------------------------- G4 schedulers
def g4_sched():
    out = {}
    cfg = dict(use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15, base_image_seq_len=256,
               max_image_seq_len=4096, shift=3.0, num_train_timesteps=1000)
    for n in (4, 30, 50):
        for S in (1152, 4096, 4736, 8192):
            mu = calculate_shift(S, 256, 4096, 0.5, 1.15)
            sig = np.linspace(1.0, 1 / n, n)
            e = FlowMatchEulerDiscreteScheduler(**cfg)
            e.set_timesteps(sigmas=sig, mu=mu)
            a = StochasticRFOvershotDiscreteScheduler(**cfg)
            a.set_timesteps(sigmas=sig, mu=mu)
            out[f"euler.n{n}.S{S}.sigmas"], out[f"euler.n{n}.S{S}.timesteps"] = e.sigmas, e.timesteps
            out[f"amo.n{n}.S{S}.sigmas"], out[f"amo.n{n}.S{S}.timesteps"] = a.sigmas, a.timesteps
            out[f"mu.S{S}"] = torch.tensor(mu, dtype=torch.float64)
    # trajectories on fixed model outputs
    n, S = 6, 4096
    mu = calculate_shift(S, 256, 4096, 0.5, 1.15)
    sig = np.linspace(1.0, 1 / n, n)
    x0 = rnd((2, 16, 64), 300)
    vs = [rnd((2, 16, 64), 301 + i) for i in range(n)]
    out["traj.x0"] = x0
    for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        e = FlowMatchEulerDiscreteScheduler(**cfg)
        e.set_timesteps(sigmas=sig, mu=mu)
        x = x0.to(dt)
        for i, t in enumerate(e.timesteps):
            x = e.step(vs[i].to(dt), t, x, return_dict=False)[0]
            out[f"traj.euler.{tag}.x{i}"] = x
        a = StochasticRFOvershotDiscreteScheduler(**cfg)
        a.set_c(2.0)
        a.set_overshot_func(lambda t, dt_: t + dt_)
        a.set_timesteps(sigmas=sig, mu=mu)
        x = x0.to(dt)
        torch.manual_seed(1234)  # step() draws eps from the GLOBAL RNG (generator never passed)
        for i, t in enumerate(a.timesteps):
            st = torch.get_rng_state()
            eps = torch.randn(x.shape, dtype=torch.float32)  # what randn_tensor will draw
            torch.set_rng_state(st)
            x, x1 = a.step(vs[i].to(dt), t, x, return_dict=False)
            out[f"traj.amo.{tag}.x{i}"], out[f"traj.amo.{tag}.x1_{i}"] = x, x1
            if tag == "f32":
                out[f"traj.amo.eps{i}"] = eps
    for i in range(n):
        out[f"traj.v{i}"] = vs[i]
    save("g4_sched", out)


# ----------------------------------------------------------------------------- G5 pipeline (latent in / latent out)
def g5_pipeline():
    from diffusers import AutoencoderKL
    out = {}
    vae = AutoencoderKL(in_channels=3, out_channels=3, block_out_channels=(8, 8, 8, 8), layers_per_block=1,
                        down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                        latent_channels=16, norm_num_groups=4, use_quant_conv=False, use_post_quant_conv=False,
                        shift_factor=0.1159, scaling_factor=0.3611)
    sched_cfg = dict(use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15, base_image_seq_len=256,
                     max_image_seq_len=4096, shift=3.0)
    H = W = 128  # -> latent 16x16 -> S = 64
    B, T = 2, 16
    lat = rnd((B, 64, 64), 400)
    mil = torch.cat([rnd((B, 64, 64), 401), (rnd((B, 64, 256), 402) > 0).float()], -1)
    pe, pooled = rnd((B, T, 64), 403), rnd((B, 32), 404)
    out["latents"], out["masked_image_latents"], out["prompt_embeds"], out["pooled"] = lat, mil, pe, pooled
    for sname in ("euler", "amo"):
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            if sname == "euler":
                sch = FlowMatchEulerDiscreteScheduler(**sched_cfg)
            else:
                sch = StochasticRFOvershotDiscreteScheduler(**sched_cfg)
                sch.set_c(2.0)
                sch.set_overshot_func(lambda t, d: t + d)
            m, _ = ref_model(G3_CFG, 7, dt)  # fresh per run: .to(bf16).to(f32) would keep bf16-rounded weights
            pipe = FluxFillPipeline(scheduler=sch, vae=vae, text_encoder=None, tokenizer=None, text_encoder_2=None,
                                    tokenizer_2=None, transformer=m)
            pipe.set_progress_bar_config(disable=True)
            steps = []
            if sname == "amo":
                torch.manual_seed(4321)
                st = torch.get_rng_state()
                eps = [torch.randn((B, 64, 64), dtype=torch.float32) for _ in range(4)]
                torch.set_rng_state(st)
                if tag == "f32":
                    for i, e in enumerate(eps):
                        out[f"amo.eps{i}"] = e

            def cb(p, i, t, kw):
                steps.append(kw["latents"].clone())
                return {}

            res = pipe(prompt_embeds=pe.to(dt), pooled_prompt_embeds=pooled.to(dt), latents=lat.to(dt),
                       masked_image_latents=mil.to(dt), height=H, width=W, num_inference_steps=4,
                       guidance_scale=30.0, output_type="latent", callback_on_step_end=cb,
                       callback_on_step_end_tensor_inputs=["latents"]).images
            out[f"{sname}.{tag}.final"] = res
            for i, s in enumerate(steps):
                out[f"{sname}.{tag}.step{i}"] = s
    save("g5_pipeline", out)


# ----------------------------------------------------------------------------- G6 layout, G7 rounding
def g6_g7():
    out = {}
    x = rnd((2, 16, 8, 12), 500)
    out["pack.in"] = x
    out["pack.out"] = FluxFillPipeline._pack_latents(x, 2, 16, 8, 12)
    out["unpack.out"] = FluxFillPipeline._unpack_latents(out["pack.out"], 64, 96, 8)
    out["ids.4x6"] = fo_ids(4, 6)
    # mask rearrangement exactly as prepare_mask_latents does it (pipeline_flux_fill.py:1563-1580)
    mask = (rnd((2, 1, 64, 96), 501) > 0).float()
    h, w = 8, 12
    mm = mask[:, 0, :, :].view(2, h, 8, w, 8).permute(0, 2, 4, 1, 3).reshape(2, 64, h, w)
    out["mask.in"] = mask
    out["mask.out"] = FluxFillPipeline._pack_latents(mm, 2, 64, h, w)
    # G7: bf16 rounding chain of t and guidance
    ts = torch.tensor([890.7682, 1000.0, 500.3, 33.3333, 7.77])
    chain = []
    for t in ts:
        a = t.expand(1).to(torch.bfloat16)          # pipeline :2082
        b = a / 1000                                # pipeline :2086
        c = b.to(torch.bfloat16) * 1000             # transformer :1088
        chain.append(c.float())
    out["round.t_in"], out["round.t_out"] = ts, torch.cat(chain)
    g = torch.full([1], 30.0, dtype=torch.float32)
    out["round.g_out"] = (g.to(torch.bfloat16) * 1000).float()
    save("g6_layout", out)


# ----------------------------------------------------------------------------- G9 VAE + VAE-inclusive pipeline
G9_VAE = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)


def g9_vae():
    from diffusers import AutoencoderKL
    from oracle import vae_oracle as vo
    out = {}
    cfg = vo.VaeConfig(**G9_VAE)
    vae = AutoencoderKL(in_channels=3, out_channels=3, block_out_channels=cfg.block_out_channels,
                        layers_per_block=cfg.layers_per_block, down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4, latent_channels=16, norm_num_groups=cfg.norm_num_groups,
                        use_quant_conv=False, use_post_quant_conv=False, shift_factor=0.1159, scaling_factor=0.3611)
    sd = vo.seeded_state_dict(cfg, 900)
    assert sorted(vae.state_dict().keys()) == sorted(sd.keys()), "oracle VAE keys != reference keys"
    vae.load_state_dict(sd, strict=True)
    vae.eval()
    x = rnd((2, 3, 64, 96), 901).clamp(-1, 1)
    out["x"] = x
    post = vae.encode(x).latent_dist
    out["enc.mean"], out["enc.std"] = post.mean, post.std
    gen = torch.Generator().manual_seed(77)
    st = gen.get_state()
    eps = torch.randn(post.mean.shape, generator=gen)
    gen.set_state(st)
    out["enc.eps"] = eps
    out["enc.sample"] = post.sample(generator=gen)
    z = rnd((2, 16, 8, 12), 902)
    out["z"] = z
    out["dec.out"] = vae.decode(z, return_dict=False)[0]
    # VAE-inclusive FluxFillPipeline.__call__: image + mask in, np image out (fp32, Euler, 3 steps)
    m, _ = ref_model(G3_CFG, 7)
    sch = FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                                          base_image_seq_len=256, max_image_seq_len=4096, shift=3.0)
    pipe = FluxFillPipeline(scheduler=sch, vae=vae, text_encoder=None, tokenizer=None, text_encoder_2=None,
                            tokenizer_2=None, transformer=m)
    pipe.set_progress_bar_config(disable=True)
    H = W = 128
    image = (rnd((2, 3, H, W), 903) * 0.3 + 0.5).clamp(0, 1)
    mask = torch.zeros(2, 1, H, W)
    mask[:, :, 32:96, 16:112] = 1.0
    pe, pooled = rnd((2, 16, 64), 904), rnd((2, 32), 905)
    gen = torch.Generator().manual_seed(4242)
    st = gen.get_state()
    lat_noise = torch.randn((2, 16, 16, 16), generator=gen)      # 1st draw: prepare_latents (pipeline :1825)
    post_eps = torch.randn((2, 16, 16, 16), generator=gen)       # 2nd draw: posterior sample (:1528)
    gen.set_state(st)
    res = pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, image=image, mask_image=mask, height=H, width=W,
               num_inference_steps=3, guidance_scale=30.0, generator=gen, output_type="np").images
    out["pipe.image"], out["pipe.mask"], out["pipe.prompt_embeds"], out["pipe.pooled"] = image, mask, pe, pooled
    out["pipe.lat_noise"], out["pipe.post_eps"] = lat_noise, post_eps
    out["pipe.out_np"] = torch.from_numpy(res)
    gen.set_state(st)
    out["pipe.out_latent"] = pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, image=image, mask_image=mask, height=H,
                                  width=W, num_inference_steps=3, guidance_scale=30.0, generator=gen,
                                  output_type="latent").images
    save("g9_vae", out)


# ----------------------------------------------------------------------------- G13 tiled VAE (enable_tiling)
def g13_vae_tiled():
    """AutoencoderKL.enable_tiling() (autoencoder_kl.py:145-160, 264-267, 301-303, 346-395, 456-503) on the G9 architecture with
    sample_size 32 (tiles of 32 px / 4 latent px every 24 / 3, blended over 1 latent px / 8 px): ragged tile grids in both directions,
    fp32 and bf16 (the blend's rounding points), encode moments and decode output."""
    from diffusers import AutoencoderKL
    from oracle import vae_oracle as vo
    out = {}
    cfg = vo.VaeConfig(**G9_VAE)
    sd = vo.seeded_state_dict(cfg, 1300)
    x = rnd((2, 3, 80, 56), 1301).clamp(-1, 1)
    z = rnd((2, 16, 10, 7), 1302)
    out["x"], out["z"] = x, z
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        vae = AutoencoderKL(in_channels=3, out_channels=3, block_out_channels=cfg.block_out_channels,
                            layers_per_block=cfg.layers_per_block, down_block_types=("DownEncoderBlock2D",) * 4,
                            up_block_types=("UpDecoderBlock2D",) * 4, latent_channels=16, norm_num_groups=cfg.norm_num_groups,
                            use_quant_conv=False, use_post_quant_conv=False, shift_factor=0.1159, scaling_factor=0.3611,
                            sample_size=32)
        vae.load_state_dict(sd, strict=True)
        vae.eval().to(dt)
        vae.enable_tiling()
        assert (vae.tile_sample_min_size, vae.tile_latent_min_size, vae.tile_overlap_factor) == (32, 4, 0.25)
        post = vae.encode(x.to(dt)).latent_dist
        out[f"{name}.enc.mean"], out[f"{name}.enc.std"] = post.mean, post.std
        out[f"{name}.dec.out"] = vae.decode(z.to(dt), return_dict=False)[0]
        vae.disable_tiling()
        out[f"{name}.dec.untiled"] = vae.decode(z.to(dt), return_dict=False)[0]
    assert not torch.equal(out["f32.dec.out"], out["f32.dec.untiled"])     # tiling changes the result (each tile its own statistics)
    save("g13_vae_tiled", out)


# ----------------------------------------------------------------------------- G15 the seam blend alone
def g15_vae_blend():
    """AutoencoderKL.blend_v / blend_h (autoencoder_kl.py:322-344) on bf16 NCHW tiles, the extent clamped by a short tile: inputs and the
    REFERENCE's outputs, so that tfx_blend_edge_nhwc is pinned to the reference's two bf16 tensor ops directly, not through the oracle."""
    from diffusers import AutoencoderKL
    vae = AutoencoderKL(in_channels=3, out_channels=3, block_out_channels=(32,), layers_per_block=1, down_block_types=("DownEncoderBlock2D",),
                        up_block_types=("UpDecoderBlock2D",), latent_channels=4, norm_num_groups=8)
    out = {}
    a = rnd((2, 16, 9, 12), 1500).to(torch.bfloat16)          # NCHW: the tile above / to the left
    b = rnd((2, 16, 5, 12), 1501).to(torch.bfloat16)          # the (shorter) tile below
    out["v.a"], out["v.b"] = a, b
    for ext in (8, 3):
        out[f"v.out.{ext}"] = vae.blend_v(a.clone(), b.clone(), ext)
    at, bt = a.transpose(2, 3).contiguous(), b.transpose(2, 3).contiguous()
    out["h.a"], out["h.b"] = at, bt
    for ext in (8, 1):
        out[f"h.out.{ext}"] = vae.blend_h(at.clone(), bt.clone(), ext)
    save("g15_vae_blend", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g9", "g13", "g15"]
    fns = dict(g1=g1_ops, g2=g2_blocks, g3=g3_model, g4=g4_sched, g5=g5_pipeline, g6=g6_g7, g9=g9_vae, g13=g13_vae_tiled, g15=g15_vae_blend)
    for w in which:
        fns[w]()
