"""Batched, multi-GPU batch driver: a list of work items -> one generated (and cropped) image per item.  Two item schemas:
the reference's `annos.json` entries {img_name, annotations: [{text, polygon}]} (scripts/run_eval.py:76-140: polygon -> mask,
strip height int(w * text_height_ratio), glyph strip stacked on top, `full_images/` + `cropped_images/` outputs) and the plain
{image, mask, text} triples of run_inference.py's rule.

Counterpart of the reference's scripts/run_eval.py:76-247.  The reference starts one worker process per GPU; every worker
holds a full replica (incl. the 9.5 GB T5 encoder), pulls ONE item at a time from a multiprocessing queue, encodes its own
prompts and calls the pipeline at batch 1.  Here (one process per GPU, launched by torchrun / `--gpus N`):

* the glyph image of every item is rendered on the host (font rasteriser); the canvas -- glyph and scene stacked, black mask
  over the glyph part, grey value of the RGB mask -- is composed ON THE DEVICE for a whole batch at once
  (`ops.compose_canvas`), including the callers' resize to a multiple of 32 (`ops.resample_u8`: Pillow's bicubic resampler
  in its own fixed-point arithmetic, bit-identical); items are grouped by pipeline geometry into batches of up to `batch_size` -- the engine's batch-8 rate is 13 % above its batch-1
  rate, and the captured step graph is reused across batches of one geometry;
* batches are dealt round-robin to the ranks.  The CLIP prompt is the one fixed template: its pooled embedding is encoded ONCE
  on rank 0 and broadcast (RCCL over xGMI).  The T5 prompts differ per image: when every rank holds a T5 encoder (9.5 GB of
  288 GB; the default) each rank encodes the prompts of ITS OWN batch -- 0.06 s per 8 prompts, no rank waits for another, the
  per-round critical path is the same on every rank.  Only when some rank has no T5 (`text_encoder_2 is None`) does rank 0
  encode for everybody and scatter, 4 MiB per prompt -- then rank 0 carries world x the encoding work on top of its own
  denoising and is the round's straggler by that much (3 % at world 8); an encoding failure on rank 0 marks that batch
  failed on its owner (a flag travels with the rows) instead of leaving the other ranks blocked in the collective;
* every rank denoises its batch with per-item generators seeded like the reference's single-image call (same noise per
  image as `run_inference.py --seed`), crops and writes its own results; a per-rank summary is gathered on rank 0.

The communication pattern (broadcast + scatter + gather, no per-step traffic) is exercised on CPU with the gloo backend
and a stub pipeline in tests/test_batch_driver_cpu.py.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import distributed as tdist
from . import glyph


@dataclass
class Work:
    index: int                       # position in the item list
    image: Any                       # composed PIL image at pipeline size (None when `parts` is set)
    mask: Any
    prompt: str                      # T5 prompt (generate_prompt(words))
    meta: Dict[str, Any]
    size: Tuple[int, int]            # (width, height) given to the pipeline
    parts: Any = None                # (glyph, scene, mask uint8 arrays, horizontal): composed on the device per batch
    name: Optional[str] = None       # eval schema: base file name written under full_images/ and cropped_images/


@dataclass
class Batch:
    size: Tuple[int, int]
    items: List[Work] = field(default_factory=list)


def eval_item_complete(item: Dict[str, Any]) -> bool:
    """The reference's filter (scripts/run_eval.py:229-231): first annotation with a non-empty text and polygon."""
    ann = item.get("annotations")
    return bool(ann) and bool(ann[0].get("text")) and bool(ann[0].get("polygon"))


def prepare_eval_item(index: int, item: Dict[str, Any], original_images_dir: str, font, text_height_ratio: float = 0.1667,
                      loader: Optional[Callable] = None, device_compose: bool = False) -> Work:
    """One `annos.json` entry -> Work (scripts/run_eval.py:76-112): scene = original_images_dir / img_name; mask = the first
    annotation's polygon filled white on black; glyph strip of height int(w * text_height_ratio) -- a fraction of the image
    WIDTH -- with the annotation's text, stacked on top with a black mask; pipeline size ((w // 32) * 32,
    ((h + strip) // 32) * 32); T5 prompt generate_prompt([text])."""
    import numpy as np
    from PIL import Image
    load = loader or (lambda p: Image.open(p))
    ann = item["annotations"][0]
    text = ann["text"]
    scene = load(os.path.join(original_images_dir, item["img_name"])).convert("RGB")
    w, h = scene.size
    strip = int(w * text_height_ratio)
    g = np.array(glyph.draw_glyph(font, text, w, strip))
    m = glyph.fill_polygon(h, w, ann["polygon"])
    meta = dict(mode="singleline", direction="vertical", strip=strip, orig_h=h)
    size = ((w // 32) * 32, ((h + strip) // 32) * 32)
    prompt = glyph.generate_prompt([text])
    name = os.path.basename(item["img_name"])
    if device_compose:
        return Work(index, None, None, prompt, meta, size, parts=(g, np.array(scene), m, False), name=name)
    combined = Image.fromarray(np.vstack((g, np.array(scene))))
    cmask = Image.fromarray(np.vstack((np.zeros_like(g), m)))
    return Work(index, combined.resize(size), cmask.resize(size), prompt, meta, size, name=name)


def prepare_item(index: int, item: Dict[str, Any], loader: Optional[Callable] = None, device_compose: bool = False,
                 eval_cfg: Optional[Dict[str, Any]] = None) -> Work:
    """Host-side preparation of one item (run_inference.py:395-467 up to the pipeline call).  With device_compose the
    stacking is left to the device when no resize is involved.  Items in the reference's `annos.json` schema (an `img_name`
    key) go through prepare_eval_item with eval_cfg = dict(original_images_dir, font, text_height_ratio)."""
    from PIL import Image
    if "img_name" in item:
        c = eval_cfg or {}
        return prepare_eval_item(index, item, c.get("original_images_dir", "."), c.get("font") or glyph.load_font(c.get("font_path")),
                                 c.get("text_height_ratio", 0.1667), loader, device_compose)
    load = loader or (lambda p: Image.open(p))
    scene, mask = load(item["image"]).convert("RGB"), load(item["mask"]).convert("RGB")
    words = glyph.read_words_from_text(item["text"])
    g, s_, m, horizontal, meta = glyph.compose_parts(scene, mask, words)
    H, W = (s_.shape[0], g.shape[1] + s_.shape[1]) if horizontal else (g.shape[0] + s_.shape[0], s_.shape[1])
    w, h = (W // 32) * 32, (H // 32) * 32
    prompt = glyph.generate_prompt(words)
    if device_compose:               # stacking, the resize to (w, h) and the grey mask happen on the device, per batch
        return Work(index, None, None, prompt, meta, (w, h), parts=(g, s_, m, horizontal))
    import numpy as np
    stack = np.hstack if horizontal else np.vstack
    combined, cmask = Image.fromarray(stack((g, s_))), Image.fromarray(stack((np.zeros_like(g), m)))
    return Work(index, combined.resize((w, h)), cmask.resize((w, h)), prompt, meta, (w, h))


def _batch_inputs(items: Sequence[Work], device):
    """(image, mask_image) arguments of the pipeline call for one batch: device-composed uint8 canvases when every item of
    the batch brought its parts and they agree in shape, else the PIL lists."""
    import numpy as np
    from . import ops
    if all(w.parts is not None for w in items):
        shapes = {(w.parts[0].shape, w.parts[1].shape, w.parts[3]) for w in items}
        if len(shapes) == 1:
            up = lambda k: torch.from_numpy(np.stack([w.parts[k] for w in items])).to(device)
            wh = items[0].size
            canvas, cmask = ops.compose_canvas(up(0), up(1), up(2), horizontal=items[0].parts[3], mask_rgb=True)
            if (canvas.shape[2], canvas.shape[1]) != wh:     # the callers' resize to a multiple of 32 (PIL bicubic, bit-exact)
                canvas, cmask = ops.resample_u8(canvas, (wh[1], wh[0])), ops.resample_u8(cmask, (wh[1], wh[0]))
            return canvas, ops.rgb_to_grey(cmask)
    from PIL import Image
    imgs, masks = [], []
    for w in items:
        if w.parts is None:
            imgs.append(w.image), masks.append(w.mask)
        else:
            g, s_, m, horizontal = w.parts
            stack = np.hstack if horizontal else np.vstack
            imgs.append(Image.fromarray(stack((g, s_))).resize(w.size))
            masks.append(Image.fromarray(stack((np.zeros_like(g), m))).resize(w.size))
    return imgs, masks


def plan_batches(works: Sequence[Work], batch_size: int) -> List[Batch]:
    """Same-geometry batches of up to batch_size items, deterministic (every rank computes the same plan): geometries in
    order of first appearance, items in list order."""
    by_size: Dict[Tuple[int, int], List[Work]] = {}
    for w in works:
        by_size.setdefault(w.size, []).append(w)
    out: List[Batch] = []
    for size, ws in by_size.items():
        for i in range(0, len(ws), batch_size):
            out.append(Batch(size, list(ws[i:i + batch_size])))
    return out


def _scatter(rows: Optional[List[torch.Tensor]], shape, dtype, device) -> torch.Tensor:
    """Rank 0 passes one tensor per rank, every rank receives its own."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return rows[0]
    out = torch.empty(shape, dtype=dtype, device=device)
    dist.scatter(out, [r.contiguous() for r in rows] if dist.get_rank() == 0 else None, src=0)
    return out


def _flag_min(v: int, device) -> int:
    """min over ranks of an int (1 rank: the value itself)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return v
    t = torch.tensor([v], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


@torch.no_grad()
def run_items(items: Sequence[Dict[str, Any]], pipe, out_dir: Optional[str], batch_size: int = 8, num_inference_steps: int = 30,
              guidance_scale: float = 30.0, seed: int = 42, device="cuda", loader: Optional[Callable] = None,
              save: Optional[Callable] = None, max_sequence_length: int = 512, eval_cfg: Optional[Dict[str, Any]] = None,
              encode: str = "auto", save_full: Optional[Callable] = None) -> Dict[str, Any]:
    """Runs the whole list; returns {"done": [indices this rank wrote], "failed": [...], "all_done": [...] on rank 0,
    "encode": "local" | "rank0"}.  `pipe` needs `encode_prompt(prompt, prompt_2, ...)` and the FluxFillPipeline `__call__`.
    encode: "local" = every rank encodes the T5 prompts of its own batches (needs a T5 on every rank), "rank0" = rank 0
    encodes for all and scatters, "auto" = local when every rank has `pipe.text_encoder_2`, else rank0.
    Outputs: eval-schema items (Work.name set) are written as out_dir/full_images/<name> and out_dir/cropped_images/<name>
    (or handed to save_full(work, full, cropped)); other items as out_dir/<index>.png (or save(index, cropped))."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    works, failed = [], []
    for i, it in enumerate(items):
        try:
            works.append(prepare_item(i, it, loader, device_compose=bool(getattr(pipe, "supports_device_compose", False)),
                                      eval_cfg=eval_cfg))
        except Exception as e:       # per-item failures do not stop the run (reference :195-198)
            failed.append(i)
            if rank == 0:
                print(f"item {i} failed in preparation: {e}")
    plan = plan_batches(works, batch_size)
    rounds = (len(plan) + world - 1) // world
    if encode == "auto":
        has_t5 = 1 if (getattr(pipe, "text_encoder_2", None) is not None or getattr(pipe, "encodes_locally", False)) else 0
        encode = "local" if _flag_min(has_t5, device) == 1 else "rank0"
    if encode not in ("local", "rank0"):
        raise ValueError("encode must be 'auto', 'local' or 'rank0'")
    # ---- the CLIP prompt is one fixed template: pooled embedding encoded once on rank 0, broadcast once.  meta[0] < 0 is the
    # abort flag: a failure on rank 0 reaches every rank through the same broadcast the others are waiting in
    pooled1 = None
    err0 = None
    if rank == 0:
        try:
            pe1, pooled1, _ = pipe.encode_prompt(prompt=glyph.PROMPT_TEMPLATE2, prompt_2=glyph.PROMPT_TEMPLATE2, device=device,
                                                 max_sequence_length=max_sequence_length)
            meta = [pe1.shape[1], pe1.shape[2], pooled1.shape[1], {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}[pe1.dtype]]
        except Exception as e:
            err0, meta = e, [-1, 0, 0, 0]
    else:
        meta = [0, 0, 0, 0]
    grouped = dist.is_initialized()      # also at world size 1 under a launcher: a 1-GPU torchrun job runs the code of the 8-GPU job
    if grouped:
        mt = torch.tensor(meta, dtype=torch.int64, device=device)
        dist.broadcast(mt, src=0)
        tdist._ran("broadcast")
        meta = [int(v) for v in mt.tolist()]
    if meta[0] < 0:
        raise RuntimeError(f"rank 0 could not encode the prompt template: {err0}" if rank == 0 else
                           "rank 0 could not encode the prompt template (see its log)")
    T, J, P, dcode = meta
    dtype = {0: torch.bfloat16, 1: torch.float32, 2: torch.float16}[dcode]
    if rank != 0:
        pooled1 = torch.empty(1, P, dtype=dtype, device=device)
    if grouped:
        dist.broadcast(pooled1, src=0)
        tdist._ran("broadcast")
    done: List[int] = []
    for r in range(rounds):
        mine = plan[r * world + rank] if r * world + rank < len(plan) else None
        pe_mine, enc_ok = None, True
        if encode == "rank0":
            # ---- rank 0 encodes the T5 prompts of every rank's batch of this round (padded to batch_size rows, one more row
            # whose first element is the ok flag) and scatters; a failed encode sends zero rows + flag 0 -- collectives stay matched
            rows = None
            if rank == 0:
                rows = []
                for k in range(world):
                    b = plan[r * world + k] if r * world + k < len(plan) else None
                    buf = torch.zeros(batch_size + 1, T, J, dtype=dtype, device=device)
                    buf[batch_size, 0, 0] = 1
                    if b is not None:
                        try:
                            prompts = [w.prompt for w in b.items]
                            pe, _, _ = pipe.encode_prompt(prompt=[glyph.PROMPT_TEMPLATE2] * len(prompts), prompt_2=prompts,
                                                          device=device, max_sequence_length=max_sequence_length)
                            buf[:len(prompts)] = pe.to(dtype)
                        except Exception as e:
                            buf[batch_size, 0, 0] = 0
                            print(f"[rank 0] encoding the prompts of batch {r * world + k} failed: {e}")
                    rows.append(buf)
            got = _scatter(rows, (batch_size + 1, T, J), dtype, device)
            pe_mine, enc_ok = got[:batch_size], bool(float(got[batch_size, 0, 0]) != 0)
        if mine is None:
            continue
        n = len(mine.items)
        try:
            if not enc_ok:
                raise RuntimeError("rank 0 failed to encode this batch's prompts")
            if encode == "local":   # this rank's own prompts, no collective: nobody waits for anybody
                prompts = [w.prompt for w in mine.items]
                pe_mine, _, _ = pipe.encode_prompt(prompt=[glyph.PROMPT_TEMPLATE2] * n, prompt_2=prompts, device=device,
                                                   max_sequence_length=max_sequence_length)
                pe_mine = pe_mine.to(dtype)
            gens = [torch.Generator(device=device).manual_seed(int(seed)) for _ in range(n)]   # run_inference.py:76, per image
            boxes = [glyph.crop_box(w.size, w.meta) for w in mine.items]
            keep_full = any(w.name is not None for w in mine.items)   # eval schema: the uncropped result is an output too
            same_box = all(bx == boxes[0] for bx in boxes)      # one crop window for the whole batch: applied on the device
            kw = dict(output_crop=boxes[0]) if same_box and not keep_full and getattr(pipe, "supports_output_crop", False) else {}
            img_in, mask_in = _batch_inputs(mine.items, device)
            images = pipe(height=mine.size[1], width=mine.size[0], image=img_in,
                          mask_image=mask_in, num_inference_steps=num_inference_steps, generator=gens,
                          max_sequence_length=max_sequence_length, guidance_scale=guidance_scale,
                          prompt_embeds=pe_mine[:n], pooled_prompt_embeds=pooled1.expand(n, -1).contiguous(), **kw).images
            for w, img, bx in zip(mine.items, images, boxes):
                cropped = img if kw else img.crop(bx)
                if w.name is not None and (save_full is not None or (save is None and out_dir is not None)):
                    if save_full is not None:
                        save_full(w, img, cropped)
                    else:
                        img.save(os.path.join(out_dir, "full_images", w.name))
                        cropped.save(os.path.join(out_dir, "cropped_images", w.name))
                elif save is not None:
                    save(w.index, cropped)
                elif out_dir is not None:
                    cropped.save(os.path.join(out_dir, f"{w.index:06d}.png"))
                done.append(w.index)
        except Exception as e:
            failed.extend(w.index for w in mine.items)
            print(f"[rank {rank}] batch of {n} at {mine.size} failed: {e}")
    # ---- summary on rank 0
    res: Dict[str, Any] = {"done": done, "failed": failed, "batches": len(plan), "rounds": rounds, "encode": encode}
    if grouped:
        cnt = torch.zeros(len(items) + 1, dtype=torch.int32, device=device)
        for i in done:
            cnt[i] = 1
        dist.all_reduce(cnt)
        tdist._ran("all_reduce")
        res["all_done"] = [i for i in range(len(items)) if int(cnt[i]) > 0]
        tdist.barrier()
    else:
        res["all_done"] = sorted(done)
    return res
