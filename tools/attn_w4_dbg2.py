import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
ops.set_option("attention_waves", 30)
def rows_of(o):
    bad = ~torch.isfinite(o.float())
    return sorted(set(bad.nonzero()[:, 1].tolist()))
for name, qs, N in [("normal", 1.0, 256), ("tinyq", 0.01, 256), ("normal", 1.0, 128), ("normal", 1.0, 192), ("bigq", 3.0, 256), ("normal", 1.0, 512)]:
    H, B = 1, 1
    Dh = 128
    y = torch.randn(B, N, 3 * Dh, device="cuda")
    y[:, :, 2 * Dh:] *= qs
    y = y.to(BF)
    q, k, v = y[:, :, 2 * Dh:], y[:, :, :Dh], y[:, :, Dh:2 * Dh]
    outs = [torch.full((B, N, Dh), 7.0, dtype=BF, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs:
        ops.attention(q, k, v, out=o)
    torch.cuda.synchronize()
    print(name, "N", N, [rows_of(o) for o in outs])
