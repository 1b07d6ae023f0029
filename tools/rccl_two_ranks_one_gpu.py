"""Experiment (round 6): can two ranks of an nccl (= RCCL) process group share the ONE GPU of a gpurun box?  The driver's scaling bench needs an
8-GPU node this builder never gets; if RCCL accepts two ranks on one device, the N = 2 branch of bench.py (broadcast of the CLIP conditioning,
all-gather of the per-rank timings, barriers) can at least execute on hardware.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_two_ranks_one_gpu.py
Prints one JSON line per rank; exit code 0 also when RCCL refuses (the refusal is the result)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
out = {"rank": rank, "world": world, "device": torch.cuda.get_device_name(0)}
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    out["all_reduce"] = x[0].item()
    y = torch.arange(8, device="cuda", dtype=torch.float32) * (rank + 1)
    dist.broadcast(y, src=0)
    g = [torch.empty(1, device="cuda") for _ in range(world)]
    dist.all_gather(g, torch.tensor([float(rank)], device="cuda"))
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    out["broadcast_ok"] = bool((y == torch.arange(8, device="cuda")).all().item())
    out["all_gather"] = [t.item() for t in g]
    out["ok"] = True
except Exception as e:  # noqa: BLE001 -- the refusal text is the finding
    out["ok"] = False
    out["error"] = f"{type(e).__name__}: {str(e)[:600]}"
print(json.dumps(out), flush=True)
try:
    dist.destroy_process_group()
except Exception:  # noqa: BLE001
    pass
sys.exit(0)
