import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
M, N, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
for v in (1, 11, 17, 26):   # full, no-prefetch, mfma-only, no-mfma
    for _ in range(2):
        ops.gemm(x, w, b, out=out, variant=v)
    torch.cuda.synchronize()
