"""Target of rocprofv3 --pmc passes: the default attention kernel at B=8, N=4608 as the DiT launches it (score bound from the norm weights:
attn_w4_kernel<4>, persistent form) and without the bound (attn_w4_kernel<0>: round 3's bookkeeping)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF); o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
for bound in (20.0, 0.0):
    for _ in range(3):
        ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o, score_bound=bound)
    torch.cuda.synchronize()
