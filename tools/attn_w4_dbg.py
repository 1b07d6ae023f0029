import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
for (B, N, H) in [(1, 64, 2), (1, 256, 2), (1, 33, 2), (1, 300, 2), (1, 1024, 1)]:
    Dh = H * 128
    y = torch.randn(B, N, 3 * Dh, device="cuda").to(BF)
    q, k, v = y[:, :, 2 * Dh:], y[:, :, :Dh], y[:, :, Dh:2 * Dh]
    ops.set_option("attention_waves", 30)
    for rep in range(2):
        o = torch.full((B, N, Dh), 7.0, dtype=BF, device="cuda")
        ops.attention(q, k, v, out=o)
        ref = torch.nn.functional.scaled_dot_product_attention(q.float().view(B, N, H, 128).transpose(1, 2), k.float().view(B, N, H, 128).transpose(1, 2),
                                                               v.float().view(B, N, H, 128).transpose(1, 2)).transpose(1, 2).reshape(B, N, Dh)
        bad = ~torch.isfinite(o.float())
        err = (o.float() - ref).abs()
        err[bad] = 0
        wrong = err > 0.05
        print(f"B{B} N{N} H{H} rep{rep}: nonfinite {int(bad.sum())}  wrong {int(wrong.sum())}  maxerr(finite) {err.max().item():.4f}")
        for name, m in (("nonfinite", bad), ("wrong", wrong)):
            if m.any():
                idx = m.nonzero()
                rows = sorted(set(idx[:, 1].tolist()))
                cols = sorted(set((idx[:, 2] % 128).tolist()))
                heads = sorted(set((idx[:, 2] // 128).tolist()))
                print(f"   {name}: heads {heads} rows[{len(rows)}] {rows[:40]} cols[{len(cols)}] {cols[:40]}")
                r0 = rows[0]
                print("   sample row", r0, o[0, r0, :16].float().tolist())
ops.set_option("attention_waves", 10)
