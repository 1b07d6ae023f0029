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
    """encode_prompt: deterministic function of the T5 prompt string.  __call__: one flat-colour image per item whose grey
    level encodes the mean of the embedding it was handed (so a mix-up of prompts is visible).
    has_t5: this rank holds a T5 encoder (`text_encoder_2`); without one only rank 0 may be asked to encode.
    slow_encode: seconds every encode_prompt call sleeps (rank 0 as a would-be straggler); fail_on: a substring of a T5
    prompt whose encoding raises."""

    def __init__(self, rank, has_t5=True, slow_encode=0.0, fail_on=None):
        self.rank, self.calls, self.encodes = rank, [], []
        self.text_encoder_2 = object() if has_t5 else None
        self.slow_encode, self.fail_on = slow_encode, fail_on

    def encode_prompt(self, prompt, prompt_2, device=None, max_sequence_length=512, **kw):
        assert self.text_encoder_2 is not None or self.rank == 0, "a rank without T5 was asked to encode"
        p2 = [prompt_2] if isinstance(prompt_2, str) else prompt_2
        if self.fail_on and any(self.fail_on in p for p in p2):
            raise RuntimeError("tokenizer exploded")
        if self.slow_encode:
            import time
            time.sleep(self.slow_encode)
        self.encodes.append(len(p2))
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


def _worker(rank, world, port, q, mode="local"):
    import time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    tdist.init_from_env(backend="gloo")
    if mode == "local":          # every rank holds T5; rank 0's encoder is SLOW: nobody else may feel it
        pipe = StubPipe(rank, has_t5=True, slow_encode=1.0 if rank == 0 else 0.0)
        real_scatter = dist.scatter
        dist.scatter = lambda *a, **k: (_ for _ in ()).throw(AssertionError("local encoding must not scatter"))
    elif mode == "rank0":        # ranks > 0 without T5: rank 0 encodes for all and scatters
        pipe = StubPipe(rank, has_t5=(rank == 0))
    else:                        # "rank0_fail": the prompts of batch 1 (owned by rank 1) cannot be encoded
        pipe = StubPipe(rank, has_t5=(rank == 0), fail_on="WORD6")
    saved, stamps = {}, []
    t0 = time.time()

    def save(i, img):
        saved[i] = (img.size, int(np.array(img)[0, 0, 0]))
        stamps.append(time.time() - t0)

    res = bd.run_items(_items(), pipe, None, batch_size=4, num_inference_steps=3, guidance_scale=30.0, seed=42, device="cpu",
                       loader=_loader, save=save)
    q.put((rank, res, saved, pipe.calls, pipe.encodes, stamps))
    dist.destroy_process_group()


def _run_world(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_two_rank_batched_driver_on_gloo():
    out = _run_world(2, "local")
    (_, r0, s0, c0, e0, t0), (_, r1, s1, c1, e1, t1) = out
    # every rank encodes its OWN batches (rank 0: the template + batches 0 and 2; rank 1: batch 1) -- no scatter happened
    # (the worker turns dist.scatter into an assertion) -- and rank 1 finished its images without ever waiting for rank 0's
    # slow encoder (3 calls x 1 s on rank 0; only the template broadcast at the start is shared)
    assert r0["encode"] == r1["encode"] == "local"
    assert e0 == [1, 4, 3] and e1 == [2]
    assert max(t1) < 2.0 and max(t0) > 2.9, (t0, t1)
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


def test_rank0_encode_mode_scatters_and_survives_an_encode_failure():
    """Ranks > 0 without a T5: rank 0 encodes for everybody and scatters (the round-2 pattern, kept for that case).  When the
    encode of a batch fails on rank 0, the owner marks the batch failed and every collective still matches (no hang)."""
    (_, r0, s0, c0, e0, _), (_, r1, s1, c1, e1, _) = _run_world(2, "rank0")
    assert r0["encode"] == r1["encode"] == "rank0" and e1 == [] and e0 == [1, 4, 2, 3]
    assert r0["all_done"] == list(range(9)) and sorted(r1["done"]) == [6, 7]
    items = _items()
    for i, (size, level) in {**s0, **s1}.items():
        words = glyph.read_words_from_text(items[i]["text"])
        assert level == int(round(float(_embed(glyph.generate_prompt(words)).mean()) * 1000)) % 256
    (_, r0, s0, c0, e0, _), (_, r1, s1, c1, e1, _) = _run_world(2, "rank0_fail")
    assert sorted(r1["failed"]) == [6, 7] and r1["done"] == [] and c1 == []
    assert sorted(r0["done"]) == [0, 1, 2, 3, 4, 5, 8] and r0["all_done"] == [0, 1, 2, 3, 4, 5, 8]


def test_world8_dry_run_every_rank_same_critical_path():
    """The 8-rank shape of `bench.py --gpus 8` / `run_eval.py --num_gpus 8`, dry (gloo, stub pipeline): 9 items in 3 batches
    over 8 ranks -- one round, ranks 3..7 idle, no rank encodes for another."""
    out = _run_world(8, "local")
    assert out[0][1]["rounds"] == 1 and out[0][1]["all_done"] == list(range(9))
    assert [o[4] for o in out] == [[1, 4]] + [[2], [3]] + [[]] * 5


def test_eval_schema_items_follow_the_reference_rule(tmp_path):
    """annos.json entries (reference scripts/run_eval.py:76-140, 168-190): strip height int(w * ratio) of the WIDTH, polygon
    mask, stacking, /32 sizes, prompt, full_images/ + cropped_images/ outputs with the reference's crop arithmetic; entries
    with incomplete annotations are not queued (:229-231)."""
    data = [dict(img_name="sub/a.png", annotations=[dict(text="HELLO", polygon=[[100, 60], [400, 60], [400, 120], [100, 120]])]),
            dict(img_name="b.png", annotations=[dict(text="", polygon=[[0, 0], [1, 1], [2, 0]])]),
            dict(img_name="c.png", annotations=[]),
            dict(img_name="d.png", annotations=[dict(text="WORLD", polygon=[[10, 10], [200, 30], [180, 90]]), dict(text="IGNORED", polygon=[[0, 0], [5, 5], [9, 0]])])]
    assert [bd.eval_item_complete(d) for d in data] == [True, False, False, True]
    scenes = {"sub/a.png": (520, 260), "b.png": (64, 64), "c.png": (64, 64), "d.png": (520, 260)}
    loader = lambda p: Image.fromarray(np.full((scenes[os.path.relpath(p, "imgs")][1], scenes[os.path.relpath(p, "imgs")][0], 3), 77, np.uint8))
    font = glyph.load_font(None)
    cfg = dict(original_images_dir="imgs", font=font, text_height_ratio=0.1667)
    w = bd.prepare_item(0, data[0], loader, eval_cfg=cfg)
    strip = int(520 * 0.1667)                                              # 86: a fraction of the WIDTH
    assert w.meta["strip"] == strip == 86 and w.meta["orig_h"] == 260 and w.name == "a.png"
    assert w.size == ((520 // 32) * 32, ((260 + 86) // 32) * 32) == (512, 320)
    assert w.image.size == (512, 320) and w.mask.size == (512, 320)
    assert w.prompt == glyph.generate_prompt(["HELLO"])
    assert glyph.crop_box(w.size, w.meta) == (0, int(320 * (86 / (260 + 86))), 512, 320) == (0, 79, 512, 320)
    m = np.array(w.mask.convert("L"))
    assert m[:70].max() == 0                                               # the glyph part's mask is black
    wd = bd.prepare_item(3, data[3], loader, device_compose=True, eval_cfg=cfg)
    g, sc, mk, horizontal = wd.parts
    assert g.shape == (86, 520, 3) and sc.shape == (260, 520, 3) and mk.shape == (260, 520, 3) and not horizontal
    assert mk[60:121, 100:401].min() == 0 and mk[20, 100, 0] == 255       # d's own triangle, not a's rectangle
    mk_a = glyph.fill_polygon(260, 520, data[0]["annotations"][0]["polygon"])
    assert mk_a[60:121, 100:401].min() == 255 and mk_a.sum() == 255 * 3 * 61 * 301   # rectangle filled inclusive of its border
    # the whole driver on the two complete items
    pipe, out = StubPipe(0), tmp_path
    os.makedirs(out / "full_images"), os.makedirs(out / "cropped_images")
    res = bd.run_items([data[0], data[3]], pipe, str(out), batch_size=8, device="cpu", loader=loader, eval_cfg=cfg)
    assert res["all_done"] == [0, 1] and pipe.calls == [(512, 320, 2)]
    for name in ("a.png", "d.png"):
        assert Image.open(out / "full_images" / name).size == (512, 320)
        assert Image.open(out / "cropped_images" / name).size == (512, 320 - 79)


def test_run_eval_cli_speaks_the_reference_flags(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import importlib
    re_ = importlib.import_module("run_eval")
    a = re_.build_parser().parse_args(["--json_path", "annos.json", "--original_images_dir", "o", "--weights_path", "w"])
    assert (a.output_dir, a.font_path, a.text_height_ratio, a.steps, a.guidance_scale, a.seed, a.num_gpus, a.scheduler) == (
        "visualization_results", "./resource/font/Arial-Unicode-Regular.ttf", 0.1667, 30, 30, 42, 4, "")
    jp = tmp_path / "annos.json"
    jp.write_text('{"data_list": [{"img_name": "x.png", "annotations": [{"text": "A", "polygon": [[0,0],[4,0],[4,4]]}]}, {"img_name": "y.png"}]}')
    lst = re_.load_data_from_json(str(jp))
    tasks, skipped = re_.select_tasks(lst)
    assert [t["img_name"] for t in tasks] == ["x.png"] and skipped == ["y.png"]
    assert re_.load_data_from_json(str(tmp_path / "missing.json")) == []
    (tmp_path / "bad.json").write_text("{nope")
    assert re_.load_data_from_json(str(tmp_path / "bad.json")) == []


def test_run_eval_lora_cli_and_format_check():
    """scripts/run_eval_lora.py (reference :148-167, :219-232): --lora_weights_path instead of --weights_path, the sampler
    defaults to "overshoot", and a file whose keys are not all LoRA / DoRA tensors is refused with the reference's message
    BEFORE anything is merged (the check sits between lora_state_dict and load_lora_into_transformer)."""
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import importlib
    re_ = importlib.import_module("run_eval")
    a = re_.build_parser(lora=True).parse_args(["--json_path", "annos.json", "--original_images_dir", "o", "--lora_weights_path", "l"])
    assert (a.lora_weights_path, a.scheduler, a.steps, a.guidance_scale, a.seed, a.num_gpus, a.text_height_ratio) == (
        "l", "overshoot", 30, 30, 42, 4, 0.1667)
    assert not hasattr(a, "weights_path")
    with pytest.raises(SystemExit):
        re_.build_parser(lora=True).parse_args(["--weights_path", "w"])           # the LoRA script has no such flag
    with pytest.raises(SystemExit):
        re_.main(["--json_path", "annos.json", "--original_images_dir", "o"], lora=True)   # --lora_weights_path is required
    merged = []

    class StubTr:
        pass

    from textflux_amd.pipeline import FluxFillPipeline
    real = FluxFillPipeline.load_lora_into_transformer
    FluxFillPipeline.load_lora_into_transformer = classmethod(lambda cls, state_dict, network_alphas, transformer, **k: merged.append(
        (sorted(state_dict), sorted(network_alphas), transformer)))
    try:
        tr = StubTr()
        good = {"transformer.x.lora_A.weight": torch.zeros(2, 4), "transformer.x.lora_B.weight": torch.zeros(4, 2),
                "transformer.x.alpha": torch.tensor(2.0)}
        assert re_.load_lora_transformer(good, base_transformer=tr) is tr
        assert merged == [(["transformer.x.lora_A.weight", "transformer.x.lora_B.weight"], ["transformer.x.alpha"], tr)]
        bad = dict(good, **{"transformer.x.weight": torch.zeros(4, 4)})
        with pytest.raises(ValueError, match="Invalid LoRA checkpoint."):
            re_.load_lora_transformer(bad, base_transformer=tr)
        assert len(merged) == 1
    finally:
        FluxFillPipeline.load_lora_into_transformer = real
    src = open(os.path.join(REPO, "scripts", "run_eval_lora.py")).read()
    assert "run_eval.main(lora=True" in src


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
