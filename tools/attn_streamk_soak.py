"""Soak of the attention kernel's stream-K tail (round 6): the same launch N times, every result compared bit for bit with the first one, on shapes
whose tail items are cut into 2 and 3 parts, both kernel forms (reference-free / guarded), interleaved with a GEMM that uses the same scratch for its
K-sliced partials (the engine lends ONE buffer to both).  usage: python tools/attn_streamk_soak.py [launches per shape]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
ops.set_option("attention_streamk", 2)
ga = torch.randn(1, 1664, 3072, device="cuda").to(BF); gw = (torch.randn(3072, 3072, device="cuda") * 0.02).to(BF); gb = torch.randn(3072, device="cuda").to(BF)
g0 = ops.gemm(ga, gw, gb, workspace=ws)         # 84 tiles < 256 CUs: K-sliced through the same scratch
bad = 0
for (B, H, N, bound) in ((8, 24, 4608, 30.0), (1, 24, 4608, 30.0), (1, 24, 1664, 30.0), (2, 5, 2304, 0.0), (3, 24, 2304, 30.0), (1, 24, 8704, 30.0)):
    D = H * 128
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
    ops.attention_mode_counts(reset=True)
    first = ops.attention(q, k, v, score_bound=bound, workspace=ws)
    assert ops.attention_mode_counts()["streamk_tail"] == 1
    diff = 0
    for i in range(n):
        if i % 7 == 0:
            diff += int(not torch.equal(ops.gemm(ga, gw, gb, workspace=ws), g0))
        diff += int(not torch.equal(ops.attention(q, k, v, score_bound=bound, workspace=ws), first))
    print(f"B {B} H {H} N {N} bound {bound}: {n} launches, {diff} differ from the first, finite {bool(torch.isfinite(first.float()).all())}", flush=True)
    bad += diff
ops.set_option("attention_streamk", 1)
print("SOAK", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
