import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
D = 3072
B, N = 8, 4608
y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
o = torch.empty(B, N, D, dtype=BF, device="cuda")
ops.set_option("attention_waves", int(sys.argv[1]) if len(sys.argv) > 1 else 30)
t = min(timeit(lambda: ops.attention(q, k, v, out=o), iters=10) for _ in range(3))
print(os.environ.get("TFX_LIB", "default")[-12:], "ms", round(t * 1e3, 4), "TF/s", round(4.0 * B * 24 * N * N * 128 / t / 1e12, 1))
