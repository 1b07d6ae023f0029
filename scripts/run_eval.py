#!/usr/bin/env python3
"""Batch driver CLI: shards a JSON list of {image, mask, text} items over the GPUs of one node.

Counterpart of the reference's scripts/run_eval.py (:76-247, flags :201-213).  The work is done by
textflux_amd/batch_driver.py: same-geometry batches (default 8 per pipeline call), prompts encoded once on rank 0 and
scattered over RCCL, every rank writes its own crops.  Defaults as the reference (30 steps, guidance 30, seed 42; strip
ratio 0.15625 as passed by batch_eval.sh).  Launch with one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_eval.py \
        --items items.json --out out/          (or:  python scripts/run_eval.py --gpus 8 ...  which re-executes itself so)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", required=True, help="JSON list of {image, mask, text} (text: string or path, lines = text lines)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--num_inference_steps", type=int, default=30)
    ap.add_argument("--guidance_scale", type=float, default=30)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--batch_size", type=int, default=8, help="same-geometry images per pipeline call")
    ap.add_argument("--gpus", type=int, default=None, help="spawn this many ranks (ignored under torchrun)")
    a = ap.parse_args()
    from textflux_amd import distributed as tdist
    tdist.respawn_under_torchrun(a.gpus, __file__, sys.argv[1:])
    rank, world, local = tdist.init_from_env()
    import run_inference as ri
    from textflux_amd import batch_driver
    with open(a.items) as f:
        items = json.load(f)
    os.makedirs(a.out, exist_ok=True)
    pipe = ri.load_flux_pipeline(text_encoders=(rank == 0))     # ranks > 0 receive their embeddings from rank 0
    pipe.enable_hip_graph(True)
    res = batch_driver.run_items(items, pipe, a.out, batch_size=a.batch_size, num_inference_steps=a.num_inference_steps,
                                 guidance_scale=a.guidance_scale, seed=a.seed, device=f"cuda:{local}")
    print(f"[rank {rank}] {len(res['done'])} images written" + (f"; {len(res['all_done'])}/{len(items)} in total, "
          f"{res['batches']} batches in {res['rounds']} rounds" if rank == 0 else ""))
    tdist.shutdown()


if __name__ == "__main__":
    main()
