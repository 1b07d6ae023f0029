"""Batch sharding of image generation over the GPUs of one node: one process per GPU, weights replicated, no
per-step communication.  The reference's only multi-GPU mechanism is independent processes fed from a
`multiprocessing` queue (scripts/run_eval.py:143-247) with every replica re-encoding its own prompts; here rank 0
encodes the conditioning once and BROADCASTS it (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm), every rank
denoises its own shard, results are GATHERED on rank 0.  Works with the `gloo` backend on CPU tensors for tests."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default process group."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # Over-subscription hooks for boxes with fewer GPUs than ranks (tests/test_world2_one_gpu.py): TFX_LOCAL_DEVICE pins every rank to
    # one device index, TFX_DIST_BACKEND names the backend.  RCCL itself refuses two ranks on one device ("Duplicate GPU detected",
    # tools/rccl_two_ranks_one_gpu.py), so the only way the world-2 branches can run with DEVICE tensors on a 1-GPU box is gloo,
    # which stages them through the host.  Neither variable is set by bench.py, the driver or scripts/run_eval.py.
    if os.environ.get("TFX_LOCAL_DEVICE", "") != "":
        local = int(os.environ["TFX_LOCAL_DEVICE"])
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # torchrun: also exercise the group at world 1
    if (world > 1 or launched) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("TFX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend not in ("nccl", "gloo"):
            raise ValueError(f"TFX_DIST_BACKEND / backend must be 'nccl' or 'gloo', got {backend!r}")
        if backend == "nccl" or (torch.cuda.is_available() and local < torch.cuda.device_count()):
            torch.cuda.set_device(local)          # nccl: fails loudly when the rank has no device of its own
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def respawn_under_torchrun(n_gpus: Optional[int], script: str, argv: Sequence[str]) -> None:
    """`python script.py --gpus N` without a launcher: replace this process by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> script.py argv`
    (one rank per GPU over RCCL).  No-op under torchrun (RANK / WORLD_SIZE set) and for N <= 1."""
    if n_gpus is None or n_gpus <= 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        return
    import socket
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(script), *argv]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes on this driver
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


# Collectives actually EXECUTED on the default group by this process, by kind (bench.py prints them: a record that says
# "rccl_ranks_seen: 1" must be able to show whether an all-reduce ran or the no-group branch answered).  Every helper below runs
# its collective whenever a group exists -- also at world size 1 under a launcher -- so that the RCCL code path of a 1-GPU
# `torch.distributed.run` job is the same code the 8-GPU job runs.
COLLECTIVES = {"all_reduce": 0, "broadcast": 0, "all_gather": 0, "gather": 0, "barrier": 0}


def _ran(kind: str) -> None:
    COLLECTIVES[kind] += 1


def group_info() -> dict:
    on = dist.is_available() and dist.is_initialized()
    return {"initialized": on, "backend": dist.get_backend() if on else None, "world_size": dist.get_world_size() if on else 1,
            "collectives_executed": dict(COLLECTIVES)}


def ranks_seen(device) -> int:
    """Number of ranks that take part in a collective on the default group (1 without a group): sum of ones."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t)
    _ran("all_reduce")
    return int(t.item())


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` work items for `rank` (first n % world ranks get one extra)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def broadcast_conditioning(prompt_embeds: Optional[torch.Tensor], pooled: Optional[torch.Tensor], shape_pe, shape_pooled,
                           dtype, device, src: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Rank `src` passes the tensors, the others pass None and receive.  Payload: 4 MiB per distinct prompt
    ([1,512,4096] bf16) + 1.5 KiB pooled -- one direct xGMI hop per peer, negligible next to >= 1 s of denoising."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank != src:
        prompt_embeds = torch.empty(shape_pe, dtype=dtype, device=device)
        pooled = torch.empty(shape_pooled, dtype=dtype, device=device)
    else:
        prompt_embeds, pooled = prompt_embeds.to(device, dtype).contiguous(), pooled.to(device, dtype).contiguous()
    if dist.is_initialized():
        dist.broadcast(prompt_embeds, src=src)
        dist.broadcast(pooled, src=src)
        _ran("broadcast"), _ran("broadcast")
    return prompt_embeds, pooled


def broadcast_tensor(t: Optional[torch.Tensor], shape, dtype, device, src: int = 0) -> torch.Tensor:
    """One tensor from rank `src` to everybody (the others pass None)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    t = torch.empty(shape, dtype=dtype, device=device) if rank != src else t.to(device, dtype).contiguous()
    if dist.is_initialized():
        dist.broadcast(t, src=src)
        _ran("broadcast")
    return t


def gather_to_rank0(t: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Equal-shape gather of per-rank results (final latents / images) on rank `dst`."""
    if not dist.is_initialized():
        return [t]
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "nccl":
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t.contiguous())   # RCCL has no rooted gather primitive cheaper than this at these sizes
        _ran("all_gather")
        return outs if rank == dst else None
    outs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t.contiguous(), outs, dst=dst)
    _ran("gather")
    return outs


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    _ran("all_reduce")
    return float(t.item())


def all_ranks(value: float, device) -> List[float]:
    """The value of every rank, in rank order (diagnostics: which rank was the straggler)."""
    if not dist.is_initialized():
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    _ran("all_gather")
    return [float(o.item()) for o in outs]


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":      # name the device: RCCL would otherwise guess it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
        _ran("barrier")
