"""Launch the dominant GEMM shapes of the P1024/B=8 workload a few times (target of the rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
for (M, N, K) in [(36864, 21504, 3072), (36864, 3072, 15360), (36864, 9216, 3072)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    for _ in range(3):
        ops.gemm(x, w, b, out=out, variant=1)
    torch.cuda.synchronize()
    print("algorithmic bytes per launch (A + W + C, bf16):", 2 * (M * K + N * K + M * N))
