#!/usr/bin/env python3
"""Batch driver: shards a JSON list of {image, mask, text} items over the GPUs of one node.

Counterpart of the reference's scripts/run_eval.py (:76-247: one process per GPU, a multiprocessing Manager queue, every
replica loads its own copy and encodes its own prompts).  Here: one process per GPU launched by torchrun, contiguous
static shards (`shard_range`), defaults as the reference (30 steps, guidance 30, seed 42, strip ratio 0.15625 as passed by
batch_eval.sh).  Launch:  python -m torch.distributed.run --nproc-per-node 8 scripts/run_eval.py --items items.json --out out/
"""
import argparse
import json
import os
import sys

import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_inference as ri
from textflux_amd import distributed as tdist
from textflux_amd import glyph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", required=True, help="JSON list of {image, mask, text} (text: string, lines = text lines)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--num_inference_steps", type=int, default=30)
    ap.add_argument("--guidance_scale", type=float, default=30)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    rank, world, local = tdist.init_from_env()
    with open(a.items) as f:
        items = json.load(f)
    os.makedirs(a.out, exist_ok=True)
    pipe = ri.load_flux_pipeline()
    done = 0
    for i in tdist.shard_range(len(items), rank, world):
        it = items[i]
        try:
            scene, mask = Image.open(it["image"]).convert("RGB"), Image.open(it["mask"]).convert("RGB")
            words = glyph.read_words_from_text(it["text"])
            combined, cmask, meta = glyph.compose(scene, mask, words)
            full = ri.run_inference(combined, cmask, it["text"], a.num_inference_steps, a.guidance_scale, a.seed, pipe=pipe)
            full.crop(glyph.crop_box(full.size, meta)).save(os.path.join(a.out, f"{i:06d}.png"))
            done += 1
        except Exception as e:  # per-item failures do not stop the shard (reference :195-198)
            print(f"[rank {rank}] item {i} failed: {e}")
    tdist.barrier()
    print(f"[rank {rank}] {done} images written")


if __name__ == "__main__":
    main()
