import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
ops.set_option("attention_waves", 0)
tot = 0
for name, qs, N, H in [("normal", 1.0, 256, 1), ("tinyq", 0.01, 256, 1), ("normal", 1.0, 128, 1), ("normal", 1.0, 192, 1), ("bigq", 3.0, 256, 1), ("normal", 1.0, 512, 1),
                       ("normal", 1.0, 1000, 4), ("bigq", 4.0, 4608, 24), ("normal", 1.0, 33, 2), ("normal", 1.0, 8704, 6)]:
    B, Dh = 1, 128 * H
    y = torch.randn(B, N, 3 * Dh, device="cuda")
    y[:, :, 2 * Dh:] *= qs
    y = y.to(BF)
    q, k, v = y[:, :, 2 * Dh:], y[:, :, :Dh], y[:, :, Dh:2 * Dh]
    outs = [torch.full((B, N, Dh), 7.0, dtype=BF, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for o in outs:
        ops.attention(q, k, v, out=o)
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float().view(B, N, H, 128).transpose(1, 2), k.float().view(B, N, H, 128).transpose(1, 2),
                                                           v.float().view(B, N, H, 128).transpose(1, 2)).transpose(1, 2).reshape(B, N, Dh)
    for i, o in enumerate(outs):
        bad = ~torch.isfinite(o.float()).all(-1)
        err = (o.float() - ref).abs()
        err[~torch.isfinite(err)] = 0
        nb = int(bad.sum())
        tot += nb + int((err > 0.05).sum())
        print(name, N, H, "run", i, "nonfinite rows", nb, bad.nonzero()[:6, 1].tolist(), "wrong elems", int((err > 0.05).sum()), "max err", round(err.max().item(), 4), "equal to run 0:", bool(torch.equal(o, outs[0])))
print("TOTAL BAD", tot)
