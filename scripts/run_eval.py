#!/usr/bin/env python3
"""Batch evaluation driver with the reference's own interface (scripts/run_eval.py:201-213 flags, :76-140 per-item rule,
:168-190 outputs), on the MI355X engine:

    python scripts/run_eval.py --json_path annos.json --original_images_dir imgs/ --weights_path models/textflux/transformer \
        [--output_dir visualization_results --font_path resource/font/Arial-Unicode-Regular.ttf --text_height_ratio 0.1667
         --steps 30 --guidance_scale 30 --seed 42 --num_gpus 4 --scheduler "" | overshoot]

`annos.json` = {"data_list": [{"img_name": ..., "annotations": [{"text": ..., "polygon": [[x, y], ...]}, ...]}, ...]}; per item
the first annotation is used: polygon -> white-on-black mask, glyph strip of height int(width * text_height_ratio) stacked on
top, resize to multiples of 32, prompt = template / generate_prompt([text]); outputs <output_dir>/full_images/<name> and
<output_dir>/cropped_images/<name> (crop top = int(res_h * strip / (orig_h + strip))).  Items with incomplete annotations
are skipped, per-item failures do not stop the run.

Where the reference starts --num_gpus worker processes that each pull ONE item at a time from a queue, this script re-executes
itself as --num_gpus ranks under torch.distributed.run (one process per GPU, RCCL) and hands the list to
textflux_amd/batch_driver.py: same-geometry batches of --batch_size (an extra flag, default 8) per pipeline call, dealt
round-robin, every rank encoding its own prompts, the fixed CLIP template broadcast once.  The pre-round-3 {image, mask, text}
list interface is still there behind --items / --out.
Model locations are local directories: --weights_path (transformer) and $TEXTFLUX_BASE (the FLUX.1-Fill-dev pipeline layout).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_data_from_json(json_path):
    """`data_list` of an annos.json; [] (with a message) when the file is missing, malformed or empty (reference :44-58)."""
    try:
        with open(json_path, "r", encoding="utf-8") as f:
            data = json.load(f)
    except FileNotFoundError:
        print(f"Error: JSON file not found at '{json_path}'.")
        return []
    except json.JSONDecodeError:
        print(f"Error: JSON file '{json_path}' has an invalid format.")
        return []
    lst = data.get("data_list", []) if isinstance(data, dict) else []
    if not lst:
        print(f"Warning: The 'data_list' in JSON file '{json_path}' is empty.")
    return lst


def build_parser(lora: bool = False):
    """lora=True: the interface of scripts/run_eval_lora.py (reference :219-232) -- --lora_weights_path instead of --weights_path,
    --scheduler defaults to "overshoot"."""
    ap = argparse.ArgumentParser(description="Batched multi-GPU stitching and FLUX-Fill inference (TextFlux evaluation driver)")
    ap.add_argument("--json_path", type=str, help="Path to the annos.json file containing annotation information")
    ap.add_argument("--original_images_dir", type=str, help="Path to the folder containing original images")
    ap.add_argument("--output_dir", type=str, default="visualization_results", help="Main output folder for results")
    if lora:
        ap.add_argument("--lora_weights_path", type=str, help="Path to lora weights")
    else:
        ap.add_argument("--weights_path", type=str, help="Path to transformer weights")
    ap.add_argument("--font_path", type=str, default="./resource/font/Arial-Unicode-Regular.ttf", help="Path to the font file (.ttf or .ttc)")
    ap.add_argument("--text_height_ratio", type=float, default=0.1667, help="Ratio of top text line height to image width (default: 1/6)")
    ap.add_argument("--steps", type=int, default=30, help="Inference steps")
    ap.add_argument("--guidance_scale", type=float, default=30, help="Guidance scale")
    ap.add_argument("--seed", type=int, default=42, help="Random seed")
    ap.add_argument("--num_gpus", type=int, default=4, help="Number of GPUs to use")
    ap.add_argument("--scheduler", type=str, default="overshoot" if lora else "", help='Sampler, None or "overshoot"')
    # not in the reference: batching, and the {image, mask, text} list interface of the earlier rounds
    ap.add_argument("--batch_size", type=int, default=8, help="same-geometry images per pipeline call")
    ap.add_argument("--items", type=str, default=None, help="JSON list of {image, mask, text} instead of --json_path")
    ap.add_argument("--out", type=str, default=None, help="output folder of --items mode")
    ap.add_argument("--num_inference_steps", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=None, help=argparse.SUPPRESS)
    return ap


def select_tasks(data_list):
    """(tasks, skipped names): the reference queues only items whose first annotation has text and polygon (:229-231)."""
    from textflux_amd import batch_driver
    tasks, skipped = [], []
    for it in data_list:
        if batch_driver.eval_item_complete(it):
            tasks.append(it)
        else:
            skipped.append(it.get("img_name"))
            print(f"Skipping {it.get('img_name')}: Incomplete annotation information.")
    return tasks, skipped


def load_lora_transformer(lora_weights_path, base_transformer=None):
    """The reference worker's load sequence (scripts/run_eval_lora.py:148-167): the BASE FLUX.1-Fill-dev transformer, the LoRA file
    through FluxFillPipeline.lora_state_dict(return_alphas=True), the format check (every key names a LoRA / DoRA tensor, else
    ValueError("Invalid LoRA checkpoint.")), then load_lora_into_transformer -- which here MERGES the update into the fused weights
    once (textflux_amd/lora.py) instead of wrapping every Linear in a PEFT layer.  `base_transformer`: an already loaded
    transformer (tests); default = $TEXTFLUX_BASE/transformer, the local copy of black-forest-labs/FLUX.1-Fill-dev/transformer."""
    import torch
    import run_inference as ri
    from textflux_amd.pipeline import FluxFillPipeline
    from textflux_amd.transformer import FluxTransformer2DModel
    transformer = base_transformer if base_transformer is not None else FluxTransformer2DModel.from_pretrained(
        ri.BASE, subfolder="transformer", torch_dtype=torch.bfloat16)
    state_dict, network_alphas = FluxFillPipeline.lora_state_dict(lora_weights_path, return_alphas=True)
    if not all("lora" in key or "dora_scale" in key for key in state_dict.keys()):
        raise ValueError("Invalid LoRA checkpoint.")
    FluxFillPipeline.load_lora_into_transformer(state_dict=state_dict, network_alphas=network_alphas, transformer=transformer)
    return transformer


def main(argv=None, lora: bool = False, script: str = __file__):
    a = build_parser(lora).parse_args(argv)
    legacy = a.items is not None
    weights = a.lora_weights_path if lora else a.weights_path
    if not legacy and not (a.json_path and a.original_images_dir and weights):
        raise SystemExit(f"--json_path, --original_images_dir and --{'lora_' if lora else ''}weights_path are required")
    from textflux_amd import distributed as tdist
    # --items (legacy) mode: --gpus only, no respawn without it (as before the reference CLI was added); the reference interface
    # defaults to --num_gpus 4 like the reference, clamped to the GPUs this host has
    if legacy:
        n_gpus = a.gpus
    else:
        n_gpus = a.gpus or a.num_gpus
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:
            have = 0
        if have and n_gpus > have:
            print(f"--num_gpus {n_gpus} > {have} visible GPUs: using {have}")
            n_gpus = have
    tdist.respawn_under_torchrun(n_gpus, script, sys.argv[1:] if argv is None else list(argv))
    rank, world, local = tdist.init_from_env()
    import run_inference as ri
    from textflux_amd import batch_driver, glyph
    steps = a.num_inference_steps or a.steps
    if legacy:
        with open(a.items) as f:
            items = json.load(f)
        out_dir, eval_cfg = a.out or a.output_dir, None
        os.makedirs(out_dir, exist_ok=True)
    else:
        data_list = load_data_from_json(a.json_path)
        if not data_list:
            print("Data list is empty, exiting program.")
            tdist.shutdown()
            return
        items, _ = select_tasks(data_list)
        out_dir = a.output_dir
        for d in (out_dir, os.path.join(out_dir, "full_images"), os.path.join(out_dir, "cropped_images")):
            os.makedirs(d, exist_ok=True)
        try:
            from PIL import ImageFont
            font = ImageFont.truetype(a.font_path, size=60)
        except (IOError, OSError):
            font = glyph.load_font(None)
            print(f"Font '{a.font_path}' not found, using default font.")
        eval_cfg = dict(original_images_dir=a.original_images_dir, font=font, text_height_ratio=a.text_height_ratio)
        if not lora:
            ri.TRANSFORMER = a.weights_path
    if lora and weights:      # every rank (the reference: every worker) loads the base transformer and merges the LoRA into it
        import torch
        from textflux_amd.pipeline import FluxFillPipeline
        pipe = FluxFillPipeline.from_pretrained(ri.BASE, transformer=load_lora_transformer(weights), torch_dtype=torch.bfloat16).to("cuda")
    else:
        pipe = ri.load_flux_pipeline(text_encoders=True)     # every rank encodes its own prompts: no rank-0 straggler
    if a.scheduler == "overshoot":
        ri.use_overshoot_sampler(pipe)
    pipe.enable_hip_graph(True)
    res = batch_driver.run_items(items, pipe, out_dir, batch_size=a.batch_size, num_inference_steps=steps,
                                 guidance_scale=a.guidance_scale, seed=a.seed, device=f"cuda:{local}", eval_cfg=eval_cfg)
    print(f"[rank {rank}] {len(res['done'])} images written" + (f"; {len(res['all_done'])}/{len(items)} in total, "
          f"{res['batches']} batches in {res['rounds']} rounds, prompts encoded {res['encode']}" if rank == 0 else ""))
    if rank == 0:
        print("[rank 0] process group: " + json.dumps(tdist.group_info()))
        print("All tasks processed.")
    tdist.shutdown()


if __name__ == "__main__":
    os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")
    main()
