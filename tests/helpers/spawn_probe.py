"""Helper of tests/test_batch_driver_cpu.py: started as a bare `python spawn_probe.py --gpus 2`, it must come back as two
gloo ranks (the mechanism `python bench.py --gpus N` / `scripts/run_eval.py --gpus N` rely on)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textflux_amd import distributed as tdist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
a = ap.parse_args()
tdist.respawn_under_torchrun(a.gpus, __file__, sys.argv[1:])
rank, world, local = tdist.init_from_env(backend="gloo")
seen = tdist.ranks_seen("cpu")
tdist.barrier()
print(f"PROBE rank {rank} world {world} ranks_seen {seen}", flush=True)
tdist.shutdown()
