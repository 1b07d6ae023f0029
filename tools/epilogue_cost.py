"""Per-tile fixed cost of the persistent GEMM: time vs K at fixed M, N (intercept = prologue/epilogue, slope = K loop)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
M, N = 36864, 12288   # 6912 tiles = 27 per CU exactly
for epi_name, epi in (("bias", ops.EPI_BIAS), ("gate_res", ops.EPI_BIAS_GATE_RES)):
    pts = []
    for K in (256, 512, 1024, 2048, 3072, 6144):
        x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
        gate = torch.randn(1, N, device="cuda").to(BF); res = torch.randn(M, N, device="cuda").to(BF)
        kw = dict(gate=gate, res=res) if epi == ops.EPI_BIAS_GATE_RES else {}
        t = timeit(lambda: ops.gemm(x, w, b, out=out, epilogue=epi, **kw), iters=20)
        xq, sx = ops.quantize_rows_fp8(x); wq, sw = ops.quantize_rows_fp8(w)
        t8 = timeit(lambda: ops.gemm_fp8(xq, sx, wq, sw, b, out=out, epilogue=epi, **kw), iters=20)
        pts.append((K, t * 1e6 / 27, t8 * 1e6 / 27))
        print(json.dumps(dict(epi=epi_name, K=K, us_per_tile_bf16=round(t * 1e6 / 27, 2), us_per_tile_fp8=round(t8 * 1e6 / 27, 2))), flush=True)
    (k0, a0, c0), (k1, a1, c1) = pts[2], pts[-1]
    sb, s8 = (a1 - a0) / (k1 - k0), (c1 - c0) / (k1 - k0)
    print(json.dumps(dict(epi=epi_name, bf16_us_per_64k=round(sb * 64, 3), bf16_fixed_us=round(a0 - sb * k0, 2),
                          fp8_us_per_128k=round(s8 * 128, 3), fp8_fixed_us=round(c0 - s8 * k0, 2))), flush=True)
