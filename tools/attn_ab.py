import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
D = 3072
for (B, N) in [(8, 4608), (1, 4608), (8, 8704), (1, 1664)]:
    y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    outs = {}
    for nw in (8, 16, 8, 16):
        ops.set_option("attention_waves", nw)
        o = torch.empty(B, N, D, dtype=BF, device="cuda")
        t = timeit(lambda: ops.attention(q, k, v, out=o), iters=10)
        outs[nw] = o
        print(json.dumps(dict(B=B, N=N, waves=nw, ms=round(t * 1e3, 4), tflops=round(4.0 * B * 24 * N * N * 128 / t / 1e12, 1))), flush=True)
    d = (outs[8].float() - outs[16].float()).abs()
    print("   max |lock-step - ping-pong| =", d.max().item(), " mean", d.mean().item(), flush=True)
ops.set_option("attention_waves", 8)
