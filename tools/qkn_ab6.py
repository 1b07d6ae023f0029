#!/usr/bin/env python3
"""Round 6, VERDICT round 5 item 1(a), GEMM side: what would moving the Q-side RMSNorm + RoPE out of the projection's epilogue (into the
attention kernel's Q prologue) return?  Sustained 3 s loops at the board's power cap (rocm-smi sampled), PRODUCT library
(tfx_gemm_bf16_qkn), alternating twice: the [k | v | q] and [k | v | q | mlp] projections bias/GELU-only, with the fused epilogue on k AND q
(what the model runs), and with it on k ONLY (an empty q range: the q tiles take the plain store path) -- the last minus the second is the
most a Q-side move could return on the GEMM side, before the attention kernel pays for the work.

    python tools/qkn_ab6.py gpurun_out/r06_qkn_ab.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops   # noqa: E402
from tools.power_profile import probe   # noqa: E402

BF = torch.bfloat16
D = 3072


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_qkn_ab.json"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    M = 36864
    nq = (1 + 0.1 * torch.randn(128, device="cuda")).to(BF)
    nk = (1 + 0.1 * torch.randn(128, device="cuda")).to(BF)
    ang = torch.randn(M, 64, device="cuda") * 3
    cs = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    # (name, N, epilogue kwargs, mode: 0 plain gemm, 1 norm on k + q, 2 norm on k only)
    cases = [("[k|v|q] bias only", 3 * D, dict(epilogue=ops.EPI_BIAS), 0),
             ("[k|v|q] norm + RoPE on k, q (the model)", 3 * D, dict(epilogue=ops.EPI_BIAS), 1),
             ("[k|v|q] norm + RoPE on k only", 3 * D, dict(epilogue=ops.EPI_BIAS), 2),
             ("[k|v|q|mlp] GELU on mlp only", 7 * D, dict(epilogue=ops.EPI_BIAS_GELU, gelu_from_col=3 * D), 0),
             ("[k|v|q|mlp] norm + RoPE on k, q + GELU (the model)", 7 * D, dict(epilogue=ops.EPI_BIAS_GELU, gelu_from_col=3 * D), 1),
             ("[k|v|q|mlp] norm + RoPE on k only + GELU", 7 * D, dict(epilogue=ops.EPI_BIAS_GELU, gelu_from_col=3 * D), 2)]
    rows = []
    x = torch.randn(M, D, device="cuda").to(BF)
    for rnd in range(2):
        for name, N, kw, mode in cases:
            w = (torch.randn(N, D, device="cuda") * 0.02).to(BF)
            b = torch.randn(N, device="cuda").to(BF)
            out = torch.empty(M, N, dtype=BF, device="cuda")
            if mode == 0:
                fn = lambda: ops.gemm(x, w, b, out=out, **kw)
            else:
                qr = (2 * D, 3 * D) if mode == 1 else (2 * D, 2 * D)
                fn = lambda: ops.gemm_qkn(x, w, b, nq, nk, cs, qr, (0, D), out=out, **kw)
            r = probe(name, fn, 2.0 * M * N * D, secs)
            r.update(round=rnd, N=N, mode=mode)
            rows.append(r)
            del w, out
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(rows, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
