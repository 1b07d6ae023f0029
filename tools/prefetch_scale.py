"""Per-CU throughput of the bare operand prefetch stream (ablation 18: no MFMA, no fragment reads) at different grid
sizes: tells a per-CU limit of the LDS-DMA path from an L2 / fabric limit."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
K = 12288
for (M, N) in [(512, 512), (1024, 1024), (2048, 2048), (4096, 4096), (8192, 8192), (36864, 3072)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    tiles = (M // 256) * (N // 256)
    for v, name in [(28, "prefetch-only"), (18, "panel0"), (1, "full"), (11, "no-prefetch")]:
        t = timeit(lambda: ops.gemm(x, w, b, out=out, variant=v), iters=10)
        rounds = -(-tiles // 256)
        per_ktile_us = t / rounds / (K // 64) * 1e6
        print(json.dumps(dict(M=M, N=N, K=K, tiles=tiles, variant=name, ms=round(t * 1e3, 4), us_per_ktile=round(per_ktile_us, 4),
                              GBps_per_cu=round(65536 / per_ktile_us / 1e3, 1))), flush=True)
