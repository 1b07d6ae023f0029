"""Driver of the rocprofv3 passes that compare the persistent GEMM with hipBLASLt on the dominant shape: a few launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
M, N, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
for _ in range(4):
    ops.gemm(x, w, b, out=out, variant=1)
torch.cuda.synchronize()
for _ in range(4):
    torch.nn.functional.linear(x, w, b)
torch.cuda.synchronize()
