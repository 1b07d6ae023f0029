"""The RCCL branch of textflux_amd/distributed.py on real hardware: bench.py launched exactly as
the driver launches the multi-GPU bench -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` -- with N = 1, so that
`init_from_env` creates an nccl (= RCCL) process group, and the conditioning broadcast, the closing barriers, the max-over-ranks
all-reduce, the per-rank all-gather and the result gather all EXECUTE on the GPU instead of being skipped by a "no group" branch.
Reference counterpart (processes + a queue, no collective): scripts/run_eval.py:143-247."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_world1_runs_every_collective_on_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--layers", "1", "2", "--steps", "1",
           "--warmup", "0", "--batch", "2", "--height", "512", "--width", "512", "--denoise-steps", "4", "--no-cpu-baseline",
           "--no-pil-delta", "--no-text-encoders"]
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["scaling"] == "weak"
    pg = rec["process_group"]
    assert pg["initialized"] and pg["backend"] == "nccl" and pg["world_size"] == 1
    assert rec["rccl_ranks_seen"] == 1            # the sum of ones of an all-reduce that RAN (next line), not the no-group answer
    ran = pg["collectives_executed"]
    assert ran["all_reduce"] >= 2 and ran["broadcast"] >= 2 and ran["all_gather"] >= 2 and ran["barrier"] >= 2, ran
    print("RCCL collectives executed at world 1:", ran)


def test_run_eval_under_torchrun_world1_runs_the_batch_drivers_collectives_on_rccl(tmp_path):
    """The PRODUCT driver under the launcher: scripts/run_eval.py (counterpart of the reference's scripts/run_eval.py:143-247 worker
    processes) as one rank of an nccl group, from a synthetic HF-layout checkpoint directory (tests/helpers/tiny_checkpoint.py): the
    template's pooled embedding and its meta row broadcast, the per-rank summary all-reduced, the closing barrier -- all on RCCL -- and
    the images written."""
    import numpy as np
    from PIL import Image
    from tests.helpers import tiny_checkpoint as tc
    root = str(tmp_path / "flux_fill_dev")
    tc.write_pipeline_dir(root)
    rng = np.random.default_rng(3)
    items = []
    for i, word in enumerate(["ALPHA", "BETA", "GAMMA"]):
        scene = Image.fromarray((rng.random((128, 256, 3)) * 255).astype(np.uint8))
        m = np.zeros((128, 256), np.uint8)
        m[32:96, 32:224] = 255
        sp, mp = str(tmp_path / f"s{i}.png"), str(tmp_path / f"m{i}.png")
        scene.save(sp)
        Image.fromarray(m).convert("RGB").save(mp)
        items.append(dict(image=sp, mask=mp, text=word))
    with open(tmp_path / "items.json", "w") as f:
        json.dump(items, f)
    out_dir = tmp_path / "out"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TEXTFLUX_BASE=root, TEXTFLUX_TRANSFORMER=os.path.join(root, "transformer"),
               TOKENIZERS_PARALLELISM="false")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "scripts", "run_eval.py"), "--items", str(tmp_path / "items.json"),
           "--out", str(out_dir), "--steps", "2", "--batch_size", "2"]
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("[rank 0] process group: ")]
    assert len(line) == 1, p.stdout[-2000:]
    pg = json.loads(line[0].split("process group: ", 1)[1])
    assert pg["initialized"] and pg["backend"] == "nccl" and pg["world_size"] == 1
    ran = pg["collectives_executed"]
    assert ran["broadcast"] >= 2 and ran["all_reduce"] >= 1 and ran["barrier"] >= 1, ran
    assert "3/3 in total" in p.stdout and "prompts encoded local" in p.stdout, p.stdout[-1500:]
    for i in range(3):
        im = Image.open(out_dir / f"{i:06d}.png")
        assert im.size[0] > 0 and np.asarray(im).std() > 0
