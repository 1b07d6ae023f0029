#!/usr/bin/env python3
"""LoRA variant of run_inference.py: base FLUX.1-Fill-dev transformer + TextFlux LoRA merged at load (reference:
run_inference_lora.py:44-73: lora_state_dict(..., return_alphas=True) + load_lora_into_transformer).  Like the
reference, --scheduler is parsed but the sampler is chosen by the module-level `scheduler_name`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import run_inference as base
from textflux_amd.pipeline import FluxFillPipeline
from textflux_amd.transformer import FluxTransformer2DModel

LORA = os.environ.get("TEXTFLUX_LORA", "./models/textflux-lora-beta")
scheduler_name = "default"


def load_flux_pipeline():
    if base.PIPE is None:
        transformer = FluxTransformer2DModel.from_pretrained(base.BASE, subfolder="transformer", torch_dtype=torch.bfloat16)
        state_dict, network_alphas = FluxFillPipeline.lora_state_dict(LORA, return_alphas=True)
        FluxFillPipeline.load_lora_into_transformer(state_dict, network_alphas, transformer)
        base.PIPE = FluxFillPipeline.from_pretrained(base.BASE, transformer=transformer, torch_dtype=torch.bfloat16).to("cuda")
    return base.PIPE


def main():
    ap = argparse.ArgumentParser(description="Flux Text Generation CLI (LoRA)")
    ap.add_argument("--image", type=str, required=True)
    ap.add_argument("--mask", type=str, required=True)
    ap.add_argument("--words", type=str, required=True)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--guidance-scale", type=float, default=30)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--scheduler", type=str, default="default", help="parsed but unused, as in the reference (:538)")
    a = ap.parse_args()
    base.scheduler_name = scheduler_name
    base.process_normal_mode(a.image, a.mask, a.words, a.steps, a.guidance_scale, a.seed, pipe=load_flux_pipeline())
    print("\nProcessing completed successfully!")


if __name__ == "__main__":
    main()
