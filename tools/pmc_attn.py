"""Target of rocprofv3 --pmc passes: the one-wave-per-SIMD attention kernels (30: 32x32x16 MFMA, 40: 16x16x32 MFMA) at B=8, N=4608."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF); o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
for nw in (30, 40):
    ops.set_option("attention_waves", nw)
    for _ in range(3):
        ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o)
    torch.cuda.synchronize()
