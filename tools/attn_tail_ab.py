#!/usr/bin/env python3
"""Tail split of the default attention kernel (the q-tiles of the last, partly filled round cut into key ranges): time per launch
with attention_tail_split 0 / 1 at the workload's shapes, and the difference of the two outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
for B, N in [(8, 4608), (1, 4608), (4, 5248), (1, 1664), (8, 8704), (2, 4608)]:
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    outs, best = [None, None], [1e9, 1e9]
    o = torch.empty(B, N, D, dtype=BF, device="cuda")
    f = lambda: ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o)
    reps = max(20, int(100 / (B * (N / 4608) ** 2)))
    for rnd in range(4):                  # alternate the settings: clock ramp must not favour one of them
        for sp in (0, 1):
            ops.set_option("attention_tail_split", sp)
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            best[sp] = min(best[sp], e0.elapsed_time(e1) / reps)
            outs[sp] = o.float().clone()
    for sp in (0, 1):
        print(f"B={B} N={N} H=24 tail_split={sp}: {best[sp]:.4f} ms  {4.0 * B * 24 * N * N * 128 / best[sp] / 1e9:.1f} TFLOP/s", flush=True)
    print("   max |diff| between the two:", (outs[0] - outs[1]).abs().max().item(), " mean |o|:", outs[0].abs().mean().item(), flush=True)
ops.set_option("attention_tail_split", 0)
