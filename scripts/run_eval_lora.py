#!/usr/bin/env python3
"""LoRA variant of scripts/run_eval.py with the reference's interface (scripts/run_eval_lora.py:219-232):

    python scripts/run_eval_lora.py --json_path annos.json --original_images_dir imgs/ --lora_weights_path models/textflux-lora-beta \
        [--output_dir ... --font_path ... --text_height_ratio 0.1667 --steps 30 --guidance_scale 30 --seed 42 --num_gpus 4
         --scheduler overshoot (the default here, as in the reference) | ""]

Every rank loads the base FLUX.1-Fill-dev transformer ($TEXTFLUX_BASE/transformer), reads the LoRA with
FluxFillPipeline.lora_state_dict(return_alphas=True), applies the reference's format check ("Invalid LoRA checkpoint.") and
load_lora_into_transformer (reference :148-167) -- merged into the fused weights at load, so a denoising step costs what it costs
without the LoRA -- then runs the same batched driver as run_eval.py (run_eval.load_lora_transformer / main(lora=True)).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import run_eval


if __name__ == "__main__":
    os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")
    run_eval.main(lora=True, script=__file__)
