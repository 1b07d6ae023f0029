#!/usr/bin/env python3
"""Cold start of the DiT: writes a synthetic FLUX.1-Fill-sized checkpoint in the HF sharded layout (bf16, 23.8 GB,
three shards + index) to a scratch directory, then times FluxTransformer2DModel.from_pretrained (loader.ShardStreamer: mmap,
pinned double-buffered staging, async H2D straight into the fused layout) against the per-tensor path it replaced
(safe_open(...).get_tensor + pageable copy_).  The files were just written, so they are read from the page cache: this is the
loader's own ceiling, not the disk's.

    python tools/loader_bench.py [--dir /dev/shm/tfx_ckpt] [--layers 19 38] [--out gpurun_out/r03_loader.json]"""
import argparse
import json
import os
import shutil
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd.transformer import FluxTransformer2DModel


def reference_shapes(cfg):
    """Key -> shape of the reference state dict, from the model's own fusion map (weights [rows, in], biases [rows])."""
    m = FluxTransformer2DModel.from_config(cfg)
    m._alloc("meta")
    out = {}
    for key, name, _ in m._fusion_map():
        rows = m._rows_of(key)
        out[key + ".weight"], out[key + ".bias"] = (rows, m.w[name + ".w"].shape[1]), (rows,)
    for key, _ in m._norm_map():
        out[key] = (128,)
    return out


def write_checkpoint(root, cfg, shard_bytes=8 << 30):
    from safetensors.torch import save_file
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "config.json"), "w") as f:
        json.dump(cfg, f)
    g = torch.Generator(device="cuda").manual_seed(1)
    shards, cur, size = [], {}, 0
    for k, shape in reference_shapes(cfg).items():
        t = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16).cpu()
        if size + t.numel() * 2 > shard_bytes and cur:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = t
        size += t.numel() * 2
    shards.append(cur)
    wmap, total = {}, 0
    for i, sh in enumerate(shards):
        fn = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(root, fn))
        for k, v in sh.items():
            wmap[k] = fn
            total += v.numel() * 2
    with open(os.path.join(root, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": wmap}, f)
    return total, len(shards)


def old_path(root):
    """The round-2 loader: whole shard -> host dict -> per-tensor synchronous pageable H2D."""
    from safetensors import safe_open
    with open(os.path.join(root, "config.json")) as f:
        m = FluxTransformer2DModel.from_config(json.load(f))
    with open(os.path.join(root, "diffusion_pytorch_model.safetensors.index.json")) as f:
        files = sorted(set(json.load(f)["weight_map"].values()))
    m._alloc("cuda")
    for fn in files:
        sd = {}
        with safe_open(os.path.join(root, fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
        m.load_state_dict(sd, strict=False, device="cuda")
    torch.cuda.synchronize()
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/dev/shm/tfx_ckpt")
    ap.add_argument("--layers", type=int, nargs=2, default=[19, 38])
    ap.add_argument("--out", default="gpurun_out/r03_loader.json")
    ap.add_argument("--sweep", action="store_true", help="also time threads x chunk x buffers combinations of the streamer")
    a = ap.parse_args()
    cfg = dict(in_channels=384, out_channels=64, num_layers=a.layers[0], num_single_layers=a.layers[1], attention_head_dim=128,
               num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
               axes_dims_rope=[16, 56, 56], patch_size=1)
    shutil.rmtree(a.dir, ignore_errors=True)
    t0 = time.time()
    total, nsh = write_checkpoint(a.dir, cfg)
    print(f"wrote {total / 1e9:.2f} GB in {nsh} shards to {a.dir} in {time.time() - t0:.0f} s", flush=True)
    torch.cuda.synchronize()
    rec = dict(bytes=total, shards=nsh, dir=a.dir, source="page cache (files just written)")
    for name, fn in (("streamer", lambda: FluxTransformer2DModel.from_pretrained(a.dir)), ("per_tensor_pageable", lambda: old_path(a.dir)),
                     ("streamer_again", lambda: FluxTransformer2DModel.from_pretrained(a.dir))):
        t0 = time.time()
        m = fn()
        torch.cuda.synchronize()
        dt = time.time() - t0
        rec[name] = dict(seconds=round(dt, 2), gb_per_s=round(total / dt / 1e9, 2))
        print(name, rec[name], flush=True)
        if name == "streamer":
            ref = {k: v.clone() for k, v in list(m.w.items())[:40]}
        if name == "per_tensor_pageable":
            assert all(torch.equal(ref[k], m.w[k]) for k in ref), "the two loaders disagree"
        del m
        torch.cuda.empty_cache()
    if a.sweep:
        rec["sweep"] = []
        for cfg_s in ("8,256,2", "16,128,3", "32,128,3", "16,64,4", "32,64,4", "24,256,2"):
            os.environ["TFX_LOADER"] = cfg_s
            t0 = time.time()
            m = FluxTransformer2DModel.from_pretrained(a.dir)
            torch.cuda.synchronize()
            dt = time.time() - t0
            rec["sweep"].append(dict(threads_chunkMiB_buffers=cfg_s, seconds=round(dt, 2), gb_per_s=round(total / dt / 1e9, 2)))
            print(rec["sweep"][-1], flush=True)
            del m
            torch.cuda.empty_cache()
        os.environ.pop("TFX_LOADER", None)
    shutil.rmtree(a.dir, ignore_errors=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
