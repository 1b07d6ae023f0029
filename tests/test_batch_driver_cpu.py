"""The batched multi-GPU driver (textflux_amd/batch_driver.py) on CPU: planning logic, and the whole
broadcast / scatter / all-reduce pattern with world_size 2 on the gloo backend around a stub pipeline."""
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from PIL import Image

from textflux_amd import batch_driver as bd
from textflux_amd import distributed as tdist
from textflux_amd import glyph

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, J, P = 6, 8, 4


def _items():
    """9 items, two geometries (512x256 scenes -> 512x320 single-line; 256x256 two-line -> 256x512), in mixed order."""
    items = []
    for i in range(9):
        if i % 3 == 2:
            items.append(dict(image=("scene", 256, 256, i), mask=("mask", 256, 256, i), text=f"AB{i}\nCD{i}"))
        else:
            items.append(dict(image=("scene", 512, 256, i), mask=("mask", 512, 256, i), text=f"WORD{i}"))
    return items


def _loader(spec):
    kind, w, h, i = spec
    if kind == "scene":
        return Image.fromarray(np.full((h, w, 3), 10 + i, np.uint8))
    m = np.zeros((h, w), np.uint8)
    m[h // 4: h // 2, w // 8: 7 * w // 8] = 255
    if h == w:
        m[5 * h // 8: 7 * h // 8, w // 4: 3 * w // 4] = 255
    return Image.fromarray(m)


def _embed(prompt: str) -> torch.Tensor:
    g = torch.Generator().manual_seed(sum(prompt.encode()) % 100003)
    return torch.randn(T, J, generator=g)


class StubPipe:
    """encode_prompt: deterministic function of the T5 prompt string, rank 0 only.  __call__: one flat-colour image per
    item whose grey level encodes the mean of the embedding it was handed (so a mix-up of prompts is visible)."""

    def __init__(self, rank):
        self.rank, self.calls = rank, []

    def encode_prompt(self, prompt, prompt_2, device=None, max_sequence_length=512, **kw):
        assert self.rank == 0, "only rank 0 may encode prompts"
        p2 = [prompt_2] if isinstance(prompt_2, str) else prompt_2
        return torch.stack([_embed(p) for p in p2]), torch.full((len(p2), P), 3.0), torch.zeros(T, 3)

    def __call__(self, height, width, image, mask_image, num_inference_steps, generator, max_sequence_length, guidance_scale,
                 prompt_embeds, pooled_prompt_embeds):
        n = len(image)
        assert prompt_embeds.shape == (n, T, J) and pooled_prompt_embeds.shape == (n, P) and len(generator) == n
        assert all(im.size == (width, height) for im in image) and torch.all(pooled_prompt_embeds == 3.0)
        assert all(g.initial_seed() == 42 for g in generator)
        self.calls.append((width, height, n))
        imgs = []
        for k in range(n):
            level = int(round(float(prompt_embeds[k].mean()) * 1000)) % 256
            imgs.append(Image.fromarray(np.full((height, width, 3), level, np.uint8)))
        return SimpleNamespace(images=imgs)


def test_plan_batches_groups_by_geometry():
    works = [bd.prepare_item(i, it, _loader) for i, it in enumerate(_items())]
    assert {w.size for w in works} == {(512, 320), (256, 512)}
    plan = bd.plan_batches(works, 4)
    assert [(b.size, [w.index for w in b.items]) for b in plan] == [
        ((512, 320), [0, 1, 3, 4]), ((512, 320), [6, 7]), ((256, 512), [2, 5, 8])]
    assert works[0].prompt == glyph.generate_prompt(["WORD0"]) and works[2].prompt == glyph.generate_prompt(["AB2", "CD2"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    tdist.init_from_env(backend="gloo")
    pipe = StubPipe(rank)
    saved = {}
    res = bd.run_items(_items(), pipe, None, batch_size=4, num_inference_steps=3, guidance_scale=30.0, seed=42, device="cpu",
                       loader=_loader, save=lambda i, img: saved.__setitem__(i, (img.size, int(np.array(img)[0, 0, 0]))))
    q.put((rank, res, saved, pipe.calls))
    dist.destroy_process_group()


def test_two_rank_batched_driver_on_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0, s0, c0), (_, r1, s1, c1) = out
    # 3 batches dealt round-robin: rank 0 gets batches 0 and 2, rank 1 batch 1; two rounds
    assert r0["batches"] == 3 and r0["rounds"] == 2
    assert sorted(r0["done"]) == [0, 1, 2, 3, 4, 5, 8] and sorted(r1["done"]) == [6, 7]
    assert r0["all_done"] == list(range(9)) and not r0["failed"] and not r1["failed"]
    assert c0 == [(512, 320, 4), (256, 512, 3)] and c1 == [(512, 320, 2)]
    # every crop has the scene's size and carries the embedding of ITS OWN prompt, wherever it was computed
    items = _items()
    for i, (size, level) in {**s0, **s1}.items():
        words = glyph.read_words_from_text(items[i]["text"])
        want = int(round(float(_embed(glyph.generate_prompt(words)).mean()) * 1000)) % 256
        assert level == want, (i, level, want)
        assert size == ((256, 256) if i % 3 == 2 else (512, 244))   # 320 - int(320 * 80 / 336): the reference's crop arithmetic


def test_single_process_driver_needs_no_process_group():
    pipe, saved = StubPipe(0), {}
    res = bd.run_items(_items()[:3], pipe, None, batch_size=8, device="cpu", loader=_loader, save=lambda i, im: saved.__setitem__(i, im.size))
    assert res["all_done"] == [0, 1, 2] and pipe.calls == [(512, 320, 2), (256, 512, 1)] and saved[2] == (256, 256)


def test_bare_command_respawns_one_rank_per_gpu():
    """`python <script> --gpus 2` with no launcher comes back as two ranks of one process group (what `python bench.py
    --gpus N` does on an N-GPU box); under torchrun / with --gpus 1 nothing is re-executed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "helpers", "spawn_probe.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    import re
    assert r.returncode == 0, r.stderr[-2000:]
    seen = sorted(re.findall(r"PROBE rank (\d) world 2 ranks_seen 2", r.stdout))      # the two ranks share one stdout pipe
    assert seen == ["0", "1"], (r.stdout, r.stderr[-2000:])
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "tests", "helpers", "spawn_probe.py")], capture_output=True,
                        text=True, timeout=120, env=env)
    assert r1.returncode == 0 and "PROBE rank 0 world 1 ranks_seen 1" in r1.stdout
