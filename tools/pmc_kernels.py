"""Tiny driver for rocprofv3 --pmc passes: a few launches of the attention and GEMM kernels at workload shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
B, N = 8, 4608
y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
o = torch.empty(B, N, D, dtype=BF, device="cuda")
for nw in (8, 16):
    ops.set_option("attention_waves", nw)
    for _ in range(2):
        ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o)
torch.cuda.synchronize()
M, Nn, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(Nn, K, device="cuda") * 0.02).to(BF)
b = torch.randn(Nn, device="cuda").to(BF); out = torch.empty(M, Nn, dtype=BF, device="cuda")
for _ in range(2):
    ops.gemm(x, w, b, out=out, variant=1)
torch.cuda.synchronize()
