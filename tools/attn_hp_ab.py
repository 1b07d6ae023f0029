"""A/B of the attention kernels on the workload's shapes: option 10 (attn_mx, tile phases) vs 20 (half-tile pipelined)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
D = 3072
for (B, N) in [(8, 4608), (1, 1664), (2, 8704), (1, 4608 - 37)]:
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    ref = None
    if B * N <= 10000:
        sl = slice(5 * 128, 6 * 128)
        ref = torch.nn.functional.scaled_dot_product_attention(q[:1, :, sl].float()[:, None], k[:1, :, sl].float()[:, None],
                                                               v[:1, :, sl].float()[:, None])[:, 0]
    outs = {}
    for nw in (10, 20, 10, 20):
        ops.set_option("attention_waves", nw)
        o = torch.empty(B, N, D, dtype=BF, device="cuda")
        t = timeit(lambda: ops.attention(q, k, v, out=o), iters=10)
        outs[nw] = o
        err = None if ref is None else (o[:1, :, 5 * 128:6 * 128].float() - ref).abs().max().item()
        print(json.dumps(dict(B=B, N=N, option=nw, ms=round(t * 1e3, 4), tflops=round(4.0 * B * 24 * N * N * 128 / t / 1e12, 1), max_err_vs_fp32=err)), flush=True)
    d = (outs[10].float() - outs[20].float()).abs()
    print("   max |mx - hp| =", d.max().item(), " mean", d.mean().item(), flush=True)
ops.set_option("attention_waves", 0)
