"""Per-workgroup fixed cost of the default attention kernel (prologue + epilogue + launch tail): time at several sequence lengths on
zero data (no power cap: cycles at the fixed 2.4 GHz clock), fitted as  t = rounds(N) * (F + tiles(N) * tau).  usage: python tools/attn_fixed_cost.py"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
B, H, D = 8, 24, 3072
rows = []
for N in (1152, 2304, 4608, 9216):
    y = torch.zeros(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    timeit(lambda: ops.attention(q, k, v, out=o, score_bound=20.0), iters=10)
    t = min(timeit(lambda: ops.attention(q, k, v, out=o, score_bound=20.0), iters=20) for _ in range(3))
    wgs = B * H * ((N + 255) // 256)
    rows.append(dict(N=N, ms=t * 1e3, wgs=wgs, rounds=math.ceil(wgs / 256), tiles=(N + 63) // 64))
    print(json.dumps(rows[-1]), flush=True)
# least squares for (F, tau) on  t / rounds = F + tiles * tau
import numpy as np
A = np.array([[1.0, r["tiles"]] for r in rows]); yv = np.array([r["ms"] * 1e3 / r["rounds"] for r in rows])
(F, tau), *_ = np.linalg.lstsq(A, yv, rcond=None)
print(json.dumps(dict(fixed_us_per_workgroup=F, us_per_tile=tau, mfma_floor_us_per_tile=72 * 32 / 2400.0)))
