import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
for (M, N, K) in [(36864, 9216, 3072), (36864, 3072, 12288)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    for v, name in [(2, "one-tile kernel"), (3, "persistent kernel"), (11, "no-prefetch"), (12, "no-ldsread"), (13, "no-prefetch,no-ldsread"), (14, "no-barrier"), (17, "mfma-only"), (18, "all-tiles-load-panel0 (L2-hot)"), (26, "no-mfma"), (28, "no-mfma,no-ldsread (pure prefetch stream)"), (27, "no-mfma,no-prefetch (LDS reads + barriers only)")]:
        t = timeit(lambda: ops.gemm(x, w, b, out=out, variant=v), iters=10)
        print(json.dumps(dict(M=M, N=N, K=K, variant=name, ms=t * 1e3, tflops=2.0 * M * N * K / t / 1e12)), flush=True)
