#!/usr/bin/env python3
"""What the fused epilogues cost in SUSTAINED operation at the board's power cap (bench library: tfx_bench_gemm_qkn attaches the q / k
RMSNorm + RoPE epilogue to plain tfx_gemm_bf16 calls): the [k | v | q] and [k | v | q | mlp] projections bias-only, with GELU, with the
norm epilogue -- 3 s loops each, alternating twice, rocm-smi sampled (clock, W, J per launch).  The s_memtime phase timers
(tools/gemm_phase_timers.py) give the same variants' CYCLES per tile; this gives their wall time once the power limit has settled.

    make -C textflux_amd/csrc bench && TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/qkn_power.py"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import _lib, ops   # noqa: E402
from tools.power_profile import probe   # noqa: E402

BF = torch.bfloat16
D = 3072


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_qkn_power.json"
    lib = _lib.lib()
    qfn = lib.tfx_bench_gemm_qkn
    qfn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    qfn.restype = None
    nw = torch.ones(128, dtype=BF, device="cuda")
    cs = torch.randn(36864, 64, 2, device="cuda")
    cases = [("[k|v|q] bias", 32768, 3 * D, 1 << 30, 0), ("[k|v|q] q/k-norm + RoPE", 32768, 3 * D, 1 << 30, 1),
             ("[k|v|q|mlp] bias", 36864, 7 * D, 1 << 30, 0), ("[k|v|q|mlp] GELU on mlp", 36864, 7 * D, 3 * D, 0),
             ("[k|v|q|mlp] q/k-norm + RoPE, GELU on mlp", 36864, 7 * D, 3 * D, 1)]
    rows = []
    for rnd in range(2):
        for name, m, N, gf, q in cases:
            x = torch.randn(m, D, device="cuda").to(BF)
            w = (torch.randn(N, D, device="cuda") * 0.02).to(BF)
            b = torch.randn(N, device="cuda").to(BF)
            out = torch.empty(m, N, dtype=BF, device="cuda")
            if q:
                qfn(nw.data_ptr(), nw.data_ptr(), cs.data_ptr(), D)
            r = probe(name, lambda: ops.gemm(x, w, b, out=out, epilogue=ops.EPI_BIAS_GELU, gelu_from_col=gf), 2.0 * m * N * D, 3.0)
            qfn(None, None, None, 0)
            r["round"] = rnd
            rows.append(r)
            del x, w, out
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(rows, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
