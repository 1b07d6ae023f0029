"""Sustained (3 s) loops of the attention kernels with rocm-smi sampling: ms, sclk, W, J per launch, on random and on zero data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.power_probe import probe
BF = torch.bfloat16
D = 3072
opts = [int(a) for a in sys.argv[1:]] or [10, 30, 8]
for data in ("random", "zero"):
    y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF) if data == "random" else torch.zeros(8, 4608, 3 * D, dtype=BF, device="cuda")
    o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
    for nw in opts:
        ops.set_option("attention_waves", nw)
        probe(f"attention option {nw}, {data} data", lambda: ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o), secs=3.0)
ops.set_option("attention_waves", 0)
