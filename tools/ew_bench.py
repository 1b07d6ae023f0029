"""Bandwidth of the HBM-bound DiT kernels at workload shapes."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
B, N, D, H, T = 8, 4608, 3072, 24, 512
y = torch.randn(B, N, 7 * D, device="cuda").to(BF)
w = [torch.randn(128, device="cuda").to(BF) for _ in range(4)]
cos, sin = torch.randn(N, 128, device="cuda"), torch.randn(N, 128, device="cuda")
t = timeit(lambda: ops.rmsnorm_rope_(y, 2 * D, 0, H, T, *w, cos, sin), iters=20)
print(json.dumps(dict(kernel="rmsnorm_rope", us=round(t * 1e6, 1), TBps=round(B * N * 2 * D * 2 * 2 / t / 1e12, 2))))
x = torch.randn(B, N, D, device="cuda").to(BF); o = torch.empty_like(x)
sh, sc = torch.randn(B, D, device="cuda").to(BF), torch.randn(B, D, device="cuda").to(BF)
t = timeit(lambda: ops.ln_modulate(x, sh, sc, out=o), iters=20)
print(json.dumps(dict(kernel="ln_modulate", us=round(t * 1e6, 1), TBps=round(B * N * D * 2 * 2 / t / 1e12, 2))))
q, s = ops.quantize_rows_fp8(x)
t = timeit(lambda: ops.quantize_rows_fp8(x, out=q, scale=s), iters=20)
print(json.dumps(dict(kernel="quantize_rows_fp8 K=3072", us=round(t * 1e6, 1), TBps=round(B * N * D * 3 / t / 1e12, 2))))
