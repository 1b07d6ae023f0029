"""Tile-order locality sweep of the persistent GEMM on the workload's five block shapes (P1024, batch 8: M = 36864 joint rows):
gemm_group_m = row tiles per group of the XCD-contiguous tile order.  Sustained timing (2 s per point: the board is
power-capped, short bursts flatter every variant), random bf16 data."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
D = 3072
SHAPES = [("qkv_mlp (single)", 36864, 7 * D, D, ops.EPI_BIAS_GELU), ("proj_out (single)", 36864, D, 5 * D, ops.EPI_BIAS_GATE_RES),
          ("qkv (double img)", 32768, 3 * D, D, ops.EPI_BIAS), ("ff1 (double img)", 32768, 4 * D, D, ops.EPI_BIAS_GELU),
          ("ff2 (double img)", 32768, D, 4 * D, ops.EPI_BIAS_GATE_RES)]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
for name, M, N, K, epi in SHAPES:
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF)
    gate = torch.randn(N, device="cuda").to(BF)
    res = torch.randn(M, N, device="cuda").to(BF) if epi == ops.EPI_BIAS_GATE_RES else None
    out = torch.empty(M, N, dtype=BF, device="cuda")
    row = {}
    for gm in (1, 2, 4, 8, 16, 32, 144):
        ops.set_option("gemm_group_m", gm)
        kw = dict(out=out, epilogue=epi, gelu_from_col=0, gate=gate if res is not None else None, res=res)
        for _ in range(3):
            ops.gemm(x, w, b, **kw)
        torch.cuda.synchronize()
        t0, n = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(20):
                ops.gemm(x, w, b, **kw)
            torch.cuda.synchronize()
            n += 20
        dt = (time.time() - t0) / n
        row[gm] = round(2.0 * M * N * K / dt / 1e12, 1)
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "tflops_by_group_m": row}), flush=True)
ops.set_option("gemm_group_m", 0)   # back to the by-shape default
