#!/usr/bin/env python3
"""Rate of the fused q / k RMSNorm + RoPE epilogue shapes of the DiT (bench library only: tfx_bench_gemm_qkn attaches the epilogue
to plain tfx_gemm_bf16 calls).  A/B two builds:  TFX_LIB=<bench .so> python tools/qkn_ab.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import _lib, ops
BF = torch.bfloat16
D = 3072
lib = _lib.lib()
qfn = lib.tfx_bench_gemm_qkn
qfn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
qfn.restype = None
nw = torch.ones(128, dtype=BF, device="cuda")
cs = torch.randn(36864, 64, 2, device="cuda")
for name, m, N, gf, q in [("[k|v|q] plain", 32768, 3 * D, 3 * D, 0), ("[k|v|q] qkn", 32768, 3 * D, 3 * D, 1),
                          ("[k|v|q|mlp] gelu", 36864, 7 * D, 3 * D, 0), ("[k|v|q|mlp] qkn+gelu", 36864, 7 * D, 3 * D, 1)]:
    x = torch.randn(m, D, device="cuda").to(BF); w = (torch.randn(N, D, device="cuda") * 0.02).to(BF); b = torch.randn(N, device="cuda").to(BF)
    out = torch.empty(m, N, dtype=BF, device="cuda")
    if q:
        qfn(nw.data_ptr(), nw.data_ptr(), cs.data_ptr(), D)
    f = lambda: ops.gemm(x, w, b, out=out, epilogue=ops.EPI_BIAS_GELU, gelu_from_col=gf)
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    qfn(None, None, None, 0)
    print(f"{name:24s} {best:.4f} ms  {2.0 * m * N * D / best / 1e9:.1f} TFLOP/s", flush=True)
    del x, w, out
