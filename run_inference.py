#!/usr/bin/env python3
"""Single-image TextFlux inference on the MI355X engine -- same CLI and function as the reference's run_inference.py
(flags --image --mask --words [--steps 30 --guidance-scale 30 --seed 42], `run_inference(image, mask, words, num_steps=50,
guidance_scale=30, seed=42)`), own implementation (reference: run_inference.py:44-106, 395-531).

Model locations are LOCAL directories (no hub access in this environment):
    TEXTFLUX_BASE  (default ./models/FLUX.1-Fill-dev)       HF pipeline layout incl. model_index.json
    TEXTFLUX_TRANSFORMER (default ./models/textflux-beta/transformer)
"""
import argparse
import os
import sys

import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from textflux_amd import glyph
from textflux_amd.pipeline import FluxFillPipeline
from textflux_amd.schedulers import StochasticRFOvershotDiscreteScheduler
from textflux_amd.transformer import FluxTransformer2DModel

scheduler_name = "default"  # "overshoot" or "default" (module-level switch, as in the reference :16)
BASE = os.environ.get("TEXTFLUX_BASE", "./models/FLUX.1-Fill-dev")
TRANSFORMER = os.environ.get("TEXTFLUX_TRANSFORMER", "./models/textflux-beta/transformer")

PIPE = None


def load_flux_pipeline(text_encoders: bool = True):
    """run_inference.py:44-57.  text_encoders=False (batch driver, ranks > 0) skips T5 / CLIP: those ranks receive their
    prompt embeddings from rank 0."""
    global PIPE
    if PIPE is None:
        transformer = FluxTransformer2DModel.from_pretrained(TRANSFORMER, torch_dtype=torch.bfloat16)
        skip = {} if text_encoders else dict(text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None)
        PIPE = FluxFillPipeline.from_pretrained(BASE, transformer=transformer, torch_dtype=torch.bfloat16, **skip).to("cuda")
    return PIPE


def use_overshoot_sampler(pipe):
    """Swap in the AMO sampler exactly as the reference does (:79-91): from_config, c = 2, overshot t + dt."""
    sch = StochasticRFOvershotDiscreteScheduler.from_config(pipe.scheduler.config)
    sch.set_c(2.0)
    sch.set_overshot_func(lambda t, dt: t + dt)
    pipe.scheduler = sch


def run_inference(image_input, mask_input, words_input, num_steps=50, guidance_scale=30, seed=42, pipe=None):
    image = (Image.open(image_input) if isinstance(image_input, str) else image_input).convert("RGB")
    mask = (Image.open(mask_input) if isinstance(mask_input, str) else mask_input).convert("RGB")
    new_w, new_h = glyph.pipe_size(image)
    image, mask = image.resize((new_w, new_h)), mask.resize((new_w, new_h))
    words = glyph.read_words_from_text(words_input) if isinstance(words_input, str) else list(words_input)
    prompt = glyph.generate_prompt(words)
    print("Generated prompt:", prompt)
    pipe = pipe or load_flux_pipeline()
    generator = torch.Generator(device="cuda").manual_seed(int(seed))
    if scheduler_name == "overshoot":
        use_overshoot_sampler(pipe)
    return pipe(height=new_h, width=new_w, image=image, mask_image=mask, num_inference_steps=num_steps,
                generator=generator, max_sequence_length=512, guidance_scale=guidance_scale,
                prompt=glyph.PROMPT_TEMPLATE2, prompt_2=prompt).images[0]


def process_normal_mode(image_path, mask_path, words_path, steps, guidance_scale, seed, pipe=None, out_dir="outputs_my"):
    scene, mask = Image.open(image_path).convert("RGB"), Image.open(mask_path).convert("RGB")
    words = glyph.read_words_from_text(words_path)
    print("Using multi-line text rendering mode" if len(words) > 1 else "Using single-line text rendering mode")
    combined, cmask, meta = glyph.compose(scene, mask, words)
    print("Starting inference...")
    full = run_inference(combined, cmask, words_path, num_steps=steps, guidance_scale=guidance_scale, seed=seed, pipe=pipe)
    cropped = full.crop(glyph.crop_box(full.size, meta))
    os.makedirs(os.path.join(out_dir, "crop"), exist_ok=True)
    n = 1
    while os.path.exists(os.path.join(out_dir, f"result_{n:04d}.png")):
        n += 1
    full.save(os.path.join(out_dir, f"result_{n:04d}.png"))
    cropped.save(os.path.join(out_dir, "crop", f"crop_{n:04d}.png"))
    print(f"\nProcessing mode: {meta['mode']}\nFull Result: {out_dir}/result_{n:04d}.png")
    return cropped


def main():
    ap = argparse.ArgumentParser(description="Flux Text Generation CLI")
    ap.add_argument("--image", type=str, required=True, help="Path to input image")
    ap.add_argument("--mask", type=str, required=True, help="Path to mask image")
    ap.add_argument("--words", type=str, required=True, help="Path to text file containing words")
    ap.add_argument("--steps", type=int, default=30, help="Number of inference steps")
    ap.add_argument("--guidance-scale", type=float, default=30, help="Guidance scale value")
    ap.add_argument("--seed", type=int, default=42, help="Random seed")
    a = ap.parse_args()
    process_normal_mode(a.image, a.mask, a.words, a.steps, a.guidance_scale, a.seed)
    print("\nProcessing completed successfully!")


if __name__ == "__main__":
    main()
