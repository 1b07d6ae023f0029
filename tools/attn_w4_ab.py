"""A/B of the attention kernels on the workload's shapes: option 10 (attn_mx, 8 waves x 32 rows) vs 30 (one wave per SIMD,
64 rows per wave), with an fp32 SDPA check of sampled heads and a ragged / tiny-N sweep."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
D = 3072
opts = [int(a) for a in sys.argv[1:]] or [10, 30]
def check(B, N, H=24):
    y = torch.randn(B, N, 3 * H * 128, device="cuda").to(BF)
    Dh = H * 128
    q, k, v = y[:, :, 2 * Dh:], y[:, :, :Dh], y[:, :, Dh:2 * Dh]
    res = {}
    for nw in opts:
        ops.set_option("attention_waves", nw)
        o = ops.attention(q, k, v)
        errs = []
        for (bi, hi) in ((0, 0), (B - 1, H - 1), (B // 2, H // 2)):
            sl = slice(hi * 128, (hi + 1) * 128)
            ref = torch.nn.functional.scaled_dot_product_attention(q[bi:bi + 1, :, sl].float()[:, None], k[bi:bi + 1, :, sl].float()[:, None],
                                                                   v[bi:bi + 1, :, sl].float()[:, None])[:, 0]
            errs.append((o[bi:bi + 1, :, sl].float() - ref).abs().max().item())
        res[nw] = (max(errs), bool(torch.isfinite(o.float()).all()))
    print(json.dumps(dict(B=B, N=N, H=H, max_err_vs_fp32={k: round(v[0], 5) for k, v in res.items()}, finite={k: v[1] for k, v in res.items()})), flush=True)
for (B, N, H) in [(1, 1, 2), (1, 8, 2), (1, 33, 2), (2, 65, 3), (1, 64, 2), (1, 129, 2), (1, 255, 2), (1, 256, 2), (1, 257, 2), (2, 300, 3), (1, 1000, 4), (1, 1664, 24), (1, 4571, 24)]:
    check(B, N, H)
for (B, N) in [(8, 4608), (1, 1664), (2, 8704)]:
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    outs = {}
    for nw in opts + opts:
        ops.set_option("attention_waves", nw)
        o = torch.empty(B, N, D, dtype=BF, device="cuda")
        t = timeit(lambda: ops.attention(q, k, v, out=o), iters=10)
        outs[nw] = o
        print(json.dumps(dict(B=B, N=N, option=nw, ms=round(t * 1e3, 4), tflops=round(4.0 * B * 24 * N * N * 128 / t / 1e12, 1))), flush=True)
    d = (outs[opts[0]].float() - outs[opts[-1]].float()).abs()
    print("   max |a - b| =", d.max().item(), " mean", d.mean().item(), flush=True)
ops.set_option("attention_waves", 0)
