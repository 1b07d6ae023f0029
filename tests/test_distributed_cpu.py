"""world_size-2 gloo tests of the multi-GPU path (batch sharding + conditioning broadcast + result gather) on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from textflux_amd import distributed as tdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = tdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    pe = torch.arange(2 * 4 * 8, dtype=torch.float32).reshape(2, 4, 8) if rank == 0 else None
    pooled = torch.full((2, 3), 7.0) if rank == 0 else None
    pe, pooled = tdist.broadcast_conditioning(pe, pooled, (2, 4, 8), (2, 3), torch.float32, "cpu")
    shard = list(tdist.shard_range(5, rank, world))
    mine = torch.full((3, 2), float(rank)) + pe.sum() * 0        # per-rank "latents"
    outs = tdist.gather_to_rank0(mine)
    mx = tdist.max_over_ranks(1.0 + rank, "cpu")
    tdist.barrier()
    q.put((rank, pe.sum().item(), pooled.sum().item(), shard, None if outs is None else [o[0, 0].item() for o in outs], mx))
    dist.destroy_process_group()


def test_two_rank_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_pe = float(sum(range(64)))
    assert res[0][1] == res[1][1] == want_pe and res[0][2] == res[1][2] == 42.0
    assert res[0][3] == [0, 1, 2] and res[1][3] == [3, 4]
    assert res[0][4] == [0.0, 1.0] and res[1][4] is None
    assert res[0][5] == res[1][5] == 2.0


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 33):
        for w in (1, 2, 8):
            got = [i for r in range(w) for i in tdist.shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(tdist.shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
