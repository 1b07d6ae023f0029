"""Energy per launch (sustained 3 s loop: mean board power x time) of the GEMM ablations: where the joules go."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.power_probe import probe
BF = torch.bfloat16
M, N, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
probe("idle-ish (tiny kernel loop)", lambda: out[:1024].zero_(), secs=3.0)
for v, name in [(3, "persistent"), (2, "one-tile full"), (11, "one-tile no-prefetch"), (12, "one-tile no-ldsread"), (13, "one-tile no-prefetch no-ldsread"),
                (17, "one-tile mfma-only"), (26, "one-tile no-mfma")]:
    probe(name, lambda: ops.gemm(x, w, b, out=out, variant=v), secs=3.0)
probe("hipBLASLt", lambda: torch.nn.functional.linear(x, w, b), secs=3.0)
D = 3072
y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF); o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
probe("attention", lambda: ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o), secs=3.0)
