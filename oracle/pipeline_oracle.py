"""CPU restatement of the FluxFillPipeline glue around the denoiser.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Citations: P = /root/reference/diffusers/src/diffusers/
pipelines/flux/pipeline_flux_fill.py (live code :1169-2137), R = /root/reference/run_inference.py.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import flux_oracle as fo
from . import sched_oracle as so

Tensor = torch.Tensor


# --------------------------------------------------------------------------- layout helpers
def pack_latents(latents: Tensor) -> Tensor:
    """_pack_latents (P:1743-1748): [B,C,h,w] -> [B,(h/2)(w/2),4C], channel-major then 2x2."""
    B, C, h, w = latents.shape
    x = latents.view(B, C, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (h // 2) * (w // 2), C * 4)


def unpack_latents(latents: Tensor, height: int, width: int, vae_scale_factor: int = 8) -> Tensor:
    """_unpack_latents (P:1752-1765): [B,S,4C] -> [B,C,h,w] with h = 2*(H // 16)."""
    B, S, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, h, w)


def latent_image_ids(h2: int, w2: int, dtype=torch.float32) -> Tensor:
    """_prepare_latent_image_ids (P:1728-1739): [h2*w2, 3] = (0, row, col)."""
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(h2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w2)[None, :]
    return ids.reshape(h2 * w2, 3).to(dtype)


def pack_mask(mask: Tensor, vae_scale_factor: int = 8) -> Tensor:
    """Mask rearrangement of prepare_mask_latents (P:1563-1580): [B,1,H,W] {0,1} ->
    [B,64,H/8,W/8] (8x8 pixel block to channels) -> pack -> [B,S,256]."""
    B = mask.shape[0]
    H, W = mask.shape[-2:]
    h, w = 2 * (H // (vae_scale_factor * 2)), 2 * (W // (vae_scale_factor * 2))
    m = mask[:, 0, :, :].view(B, h, vae_scale_factor, w, vae_scale_factor).permute(0, 2, 4, 1, 3)
    m = m.reshape(B, vae_scale_factor * vae_scale_factor, h, w)
    return pack_latents(m)


def masked_image_latents_from(z: Tensor, mask: Tensor, shift_factor: float, scaling_factor: float,
                              dtype=None) -> Tensor:
    """(z - shift) * scale -> pack -> cat with packed mask (P:1530-1531, 1554-1560, 2046)."""
    z = (z - shift_factor) * scaling_factor
    if dtype is not None:
        z = z.to(dtype)
    zp = pack_latents(z)
    mp = pack_mask(mask).to(zp.dtype)
    return torch.cat((zp, mp), dim=-1)


# --------------------------------------------------------------------------- bf16 rounding chain of t / guidance
def timestep_chain(t: Tensor, dtype) -> Tensor:
    """What the transformer's sinusoid finally sees for scheduler timestep `t` (f32 scalar tensor):
    t.expand(B).to(dtype) (P:2082) -> /1000 (P:2086) -> .to(dtype)*1000 (transformer_flux.py:1088).
    In bf16 890.77 becomes 892.0."""
    ts = t.reshape(1).to(dtype)
    return ((ts / 1000).to(dtype) * 1000).float()


def guidance_chain(g: float, dtype) -> Tensor:
    """guidance = full([1], g, f32) (P:2070) -> .to(dtype)*1000 (transformer_flux.py:1090): 30 -> 29952 in bf16."""
    return (torch.full([1], g, dtype=torch.float32).to(dtype) * 1000).float()


# --------------------------------------------------------------------------- denoise loop (latent -> latent)
def denoise(
    sd: fo.SD,
    cfg: fo.FluxConfig,
    latents: Tensor,                # [B,S,64] packed noise
    masked_image_latents: Tensor,   # [B,S,320]
    prompt_embeds: Tensor,          # [B,T,J]
    pooled: Tensor,                 # [B,P]
    h2: int, w2: int,               # latent grid after 2x2 packing
    num_inference_steps: int,
    guidance_scale: float,
    scheduler: str = "euler",       # "euler" | "amo"
    amo_noise: Optional[Sequence[Tensor]] = None,
    amo_c: float = 2.0,
    sched_cfg: Optional[dict] = None,
    model_fn: Optional[Callable] = None,
    max_steps: Optional[int] = None,   # stop after this many steps of the num_inference_steps schedule (full-size tests)
    teacher: Optional[Sequence[Tensor]] = None,   # teacher forcing: step i > 0 starts from teacher[i - 1] instead of its own result
) -> Tuple[Tensor, List[Tensor]]:
    """Steps 4-7 of FluxFillPipeline.__call__ with `latents=`/`masked_image_latents=`/`prompt_embeds=` injected
    and output_type='latent' (P:2012-2116).  Returns (final latents, per-step latents)."""
    sc = dict(base_image_seq_len=256, max_image_seq_len=4096, base_shift=0.5, max_shift=1.15)
    sc.update(sched_cfg or {})
    B, S, _ = latents.shape
    dtype = prompt_embeds.dtype
    latents = latents.to(dtype)
    img_ids = latent_image_ids(h2, w2, dtype)
    txt_ids = torch.zeros(prompt_embeds.shape[1], 3, dtype=dtype)
    mu = so.calculate_shift(S, sc["base_image_seq_len"], sc["max_image_seq_len"], sc["base_shift"], sc["max_shift"])
    lin = so.pipeline_sigmas(num_inference_steps)
    sig = so.euler_sigmas(lin, mu) if scheduler == "euler" else so.amo_sigmas(lin, mu)
    timesteps = so.timesteps_from_sigmas(sig)
    guidance = torch.full([1], guidance_scale, dtype=torch.float32).expand(B) if cfg.guidance_embeds else None
    fwd = model_fn or (lambda **kw: fo.transformer_forward(sd, cfg, **kw))
    traj = []
    for i, t in enumerate(timesteps):
        if max_steps is not None and i >= max_steps:
            break
        if teacher is not None and i > 0:
            latents = teacher[i - 1].to(dtype).reshape(latents.shape)
        timestep = t.expand(B).to(latents.dtype)
        noise_pred = fwd(
            hidden_states=torch.cat((latents, masked_image_latents.to(latents.dtype)), dim=2),
            timestep=timestep / 1000, guidance=guidance, pooled_projections=pooled,
            encoder_hidden_states=prompt_embeds, txt_ids=txt_ids, img_ids=img_ids)
        if scheduler == "euler":
            latents = so.euler_step(noise_pred, latents, sig[i], sig[i + 1])
        else:
            latents, _ = so.amo_step(noise_pred, latents, sig[i], sig[i + 1], amo_noise[i], amo_c)
        traj.append(latents)
    return latents, traj


# --------------------------------------------------------------------------- driver-level geometry (a19)
PROMPT_TEMPLATE2 = (
    "The pair of images highlights some white words on a black background, as well as their style on a "
    "real-world scene image. [IMAGE1] is a template image rendering the text, with the words; [IMAGE2] shows "
    "the text content naturally and correspondingly integrated into the image."
)


def generate_prompt(words: Sequence[str]) -> str:
    """generate_prompt (R:27-34)."""
    w = ", ".join(f"'{x}'" for x in words)
    return (
        "The pair of images highlights some white words on a black background, as well as their style on a "
        f"real-world scene image. [IMAGE1] is a template image rendering the text, with the words {w}; [IMAGE2] "
        f"shows the text content {w} naturally and correspondingly integrated into the image."
    )


def driver_geometry(scene_w: int, scene_h: int, multiline: bool, strip_ratio: float = 0.15625):
    """Geometry known answers of SURVEY.md Appendix F, from R:65-69 (floor to /32), R:163-165 (strip height
    int(w*ratio)), R:378-384 (horizontal iff h > w), R:409-467 (concat order, crop arithmetic).
    Returns dict(concat=(W,H), pipe=(W,H), S=tokens, crop=(l,t,r,b), direction=...)."""
    if multiline:
        direction = "horizontal" if scene_h > scene_w else "vertical"
        cw, ch = (2 * scene_w, scene_h) if direction == "horizontal" else (scene_w, 2 * scene_h)
        strip = None
    else:
        direction = "vertical"
        strip = int(scene_w * strip_ratio)
        cw, ch = scene_w, scene_h + strip
    pw, ph = (cw // 32) * 32, (ch // 32) * 32
    if multiline:
        crop = (pw // 2, 0, pw, ph) if direction == "horizontal" else (0, ph // 2, pw, ph)
    else:
        crop = (0, int(ph * (strip / (scene_h + strip))), pw, ph)
    return dict(concat=(cw, ch), pipe=(pw, ph), S=(ph // 16) * (pw // 16), crop=crop, direction=direction, strip=strip)


# --------------------------------------------------------------------------- full Fill pipeline (image + mask in)
def fill_pipeline(sd, cfg, vae_sd, vae_cfg, image: Tensor, mask: Tensor, prompt_embeds: Tensor, pooled: Tensor,
                  lat_noise: Tensor, post_eps: Tensor, num_inference_steps: int, guidance_scale: float,
                  scheduler: str = "euler", amo_noise=None, output_type: str = "np"):
    """FluxFillPipeline.__call__ with tensor image [B,3,H,W] in [0,1] and mask [B,1,H,W] (P:1963-2137), the two RNG
    draws injected (lat_noise: prepare_latents P:1825; post_eps: posterior sample P:1528, in that order -- SURVEY
    Appendix E).  Preprocess = VaeImageProcessor on tensors: normalize to [-1,1]; mask binarize at 0.5
    (D/image_processor.py:535-536, 700-716)."""
    from . import vae_oracle as vo
    dtype = prompt_embeds.dtype
    B, _, H, W = image.shape
    latents = pack_latents(lat_noise.to(dtype))
    img = 2.0 * image - 1.0
    m = mask.clone()
    m[m < 0.5] = 0
    m[m >= 0.5] = 1
    masked = (img * (1 - m)).to(dtype)
    mean, std = vo.encode_moments(masked, vae_sd, vae_cfg)
    z = vo.sample_posterior(mean, std, post_eps.to(dtype))
    mil = masked_image_latents_from(z, m, vae_cfg.shift_factor, vae_cfg.scaling_factor, dtype)
    h2, w2 = H // 16, W // 16
    final, traj = denoise(sd, cfg, latents, mil, prompt_embeds, pooled, h2, w2, num_inference_steps, guidance_scale,
                          scheduler, amo_noise)
    if output_type == "latent":
        return final
    zz = unpack_latents(final, H, W) / vae_cfg.scaling_factor + vae_cfg.shift_factor
    dec = vo.decoder(zz, vae_sd, vae_cfg)
    out = (dec / 2 + 0.5).clamp(0, 1)
    return out.cpu().permute(0, 2, 3, 1).float()
