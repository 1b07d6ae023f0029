import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
for zeros in (False, True):
    for (M, N, K) in [(36864, 9216, 3072), (2048, 2048, 12288)]:
        x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        if zeros: x.zero_(); w.zero_()
        b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
        for v, name in [(1, "full"), (11, "no-prefetch"), (10 + 32768, "half-bytes"), (10 + 65536, "L1-hot source"), (10 + 98304, "half + L1-hot"), (18, "panel0")]:
            t = timeit(lambda: ops.gemm(x, w, b, out=out, variant=v), iters=20)
            print(json.dumps(dict(zeros=zeros, M=M, N=N, K=K, variant=name, ms=round(t * 1e3, 4), tflops=round(2.0 * M * N * K / t / 1e12, 1))), flush=True)
