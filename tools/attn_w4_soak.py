"""Soak test of the default attention kernel: many back-to-back launches at the workload's shapes (interleaved with GEMMs and
the 8-wave kernel, which leave other register / LDS contents behind) must be bit-identical to the first and finite."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (B, N) in [(8, 4608), (2, 8704), (1, 1664), (4, 5248), (1, 4571)]:
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    w = (torch.randn(D, D, device="cuda") * 0.02).to(BF)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    ops.set_option("attention_waves", 10)
    ref10 = ops.attention(q, k, v)
    ops.set_option("attention_waves", 0)
    first = ops.attention(q, k, v)
    d = (first.float() - ref10.float()).abs().max().item()
    mism = 0
    for i in range(n):
        if i % 3 == 1:
            ops.gemm(y[0, :2048, :D].contiguous(), w)
        if i % 7 == 3:
            ops.set_option("attention_waves", 10)
            ops.attention(q, k, v)
            ops.set_option("attention_waves", 0)
        o = ops.attention(q, k, v)
        if not torch.equal(o, first):
            mism += 1
    fin = bool(torch.isfinite(first.float()).all())
    print(f"B {B} N {N}: {n} launches, {mism} differ from the first, finite {fin}, max |kernel 30 - kernel 10| = {d:.4f}", flush=True)
    bad += mism + (0 if fin else 1)
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
