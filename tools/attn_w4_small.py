import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
ops.set_option("attention_waves", 0)
for N in (8, 33, 64, 128, 192, 256):
    H, B, Dh = 1, 1, 128
    y = torch.randn(B, N, 3 * Dh, device="cuda").to(BF)
    q, k, v = y[:, :, 2 * Dh:], y[:, :, :Dh], y[:, :, Dh:2 * Dh]
    o = ops.attention(q, k, v)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    err = (o.float() - ref).abs()
    bad = (err > 0.05) | ~torch.isfinite(err)
    rows = sorted(set(bad.nonzero()[:, 1].tolist()))
    cols = sorted(set(bad.nonzero()[:, 2].tolist()))
    print(os.environ.get("TFX_LIB", "default")[-8:], "N", N, "bad elems", int(bad.sum()), "rows", rows[:20], "cols", cols[:20], len(cols))
