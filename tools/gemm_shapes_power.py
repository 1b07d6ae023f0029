#!/usr/bin/env python3
"""Sustained (power-capped steady state) rate of the persistent GEMM on every GEMM shape of the P1024 / batch-8 DiT step, EACH WITH THE
EPILOGUE IT CARRIES IN THE MODEL; rocm-smi sampled alongside.  Round 6: the two projections that carry the
fused q / k RMSNorm + RoPE epilogue in the model -- 47 % of a forward's GEMM FLOPs -- are timed WITH it (tfx_gemm_bf16_qkn, the product
library's own entry point; rounds 3-5 timed them bias-only / GELU-only, which flattered the comparison), on the row counts the model
launches (the text and image projections of a double block are one 36864-row launch).  --hipblaslt adds, per shape, hipBLASLt with the
only epilogue torch exposes (bias) AND the separate passes it would then need to deliver what our launch delivers: tfx_rmsnorm_rope over
the q / k columns (our own HBM-bound pass, 4.8 TB/s), torch's GELU over the mlp columns, the gated residual as torch ops -- "fair" =
hipBLASLt + those passes, each timed in its own sustained loop.

    python tools/gemm_shapes_power.py --tag r06 --hipblaslt --out gpurun_out/r06_gemm_shapes.jsonl"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops  # noqa: E402
from tools.power_profile import probe  # noqa: E402

BF = torch.bfloat16
D = 3072


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--secs", type=float, default=2.5)
    ap.add_argument("--out", default="gpurun_out/r06_gemm_shapes.jsonl")
    ap.add_argument("--hipblaslt", action="store_true")
    ap.add_argument("--plain-qkv", action="store_true", help="rounds 3-5 behaviour: the q|k|v projections WITHOUT their norm + RoPE epilogue")
    ap.add_argument("--place", type=int, default=0, help="tfx_set_option gemm_place (bench knob)")
    ap.add_argument("--opt", action="append", default=[], help="name=value for tfx_set_option (bench knobs)")
    a = ap.parse_args()
    for kv in a.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if a.place:
        ops.set_option("gemm_place", a.place)
    M = 36864          # 8 x 4608 token rows (text + image: the double block's projections run as ONE row-split launch over them)
    # (name, rows, N, K, epilogue, fused q/k norm, FLOP share of a P1024 forward in units of D^2 per token)
    shapes = [("double qkv (txt+img, q/k-norm + RoPE)", M, 3 * D, D, ops.EPI_BIAS, True, 19 * 3), ("double ff1 (GELU)", M, 4 * D, D, ops.EPI_BIAS_GELU, False, 19 * 4),
              ("double ff2 (gate + residual)", M, D, 4 * D, ops.EPI_BIAS_GATE_RES, False, 19 * 4), ("double out (gate + residual)", M, D, D, ops.EPI_BIAS_GATE_RES, False, 19 * 1),
              ("single qkv|mlp (q/k-norm + RoPE, GELU on mlp)", M, 7 * D, D, ops.EPI_BIAS_GELU, True, 38 * 7), ("single proj_out (gate + residual)", M, D, 5 * D, ops.EPI_BIAS_GATE_RES, False, 38 * 5)]
    nq = (1 + 0.1 * torch.randn(128, device="cuda")).to(BF)
    nk = (1 + 0.1 * torch.randn(128, device="cuda")).to(BF)
    ang = torch.randn(M, 64, device="cuda") * 3
    cs = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous()
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
    rows = []
    for name, m, N, K, epi, qkn, share in shapes:
        x = torch.randn(m, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        b = torch.randn(N, device="cuda").to(BF)
        out = torch.empty(m, N, dtype=BF, device="cuda")
        gate = torch.randn(1, N, device="cuda").to(BF)
        res = torch.randn(m, N, device="cuda").to(BF)
        kw = dict(epilogue=epi)
        if epi == ops.EPI_BIAS_GELU:
            kw["gelu_from_col"] = 3 * D if N == 7 * D else 0
        if epi == ops.EPI_BIAS_GATE_RES:
            kw.update(gate=gate, res=res)
        fl = 2.0 * m * N * K
        if qkn and not a.plain_qkv:
            fn = lambda: ops.gemm_qkn(x, w, b, nq, nk, cs, (2 * D, 3 * D), (0, D), out=out, **kw)
        else:
            fn = lambda: ops.gemm(x, w, b, out=out, **kw)
        r = probe(f"{a.tag} {name} {m}x{N}x{K}", fn, fl, a.secs)
        r.update(tag=a.tag, shape=name, M=m, N=N, K=K, share=share, fused_qk_norm=bool(qkn and not a.plain_qkv))
        rows.append(r)
        if a.hipblaslt:
            r2 = probe(f"hipBLASLt (bias only) {name} {m}x{N}x{K}", lambda: torch.nn.functional.linear(x, w, b), fl, a.secs)
            r2.update(tag="hipblaslt", shape=name, M=m, N=N, K=K, share=share)
            # the passes hipBLASLt's bias-only result still needs to become what our launch stores
            extra = []
            y = torch.nn.functional.linear(x, w, b)
            if qkn:
                y3 = y.view(1, m, N)
                e = probe(f"  + tfx_rmsnorm_rope over q, k of {name}", lambda: ops.rmsnorm_rope_(y3, 2 * D, 0, 24, 0, nq, nk, nq, nk, cos, sin), 0, 1.5)
                extra.append(dict(what="tfx_rmsnorm_rope (q, k columns in place)", ms=e["ms_per_launch"]))
            if epi == ops.EPI_BIAS_GELU:
                lo = kw["gelu_from_col"]
                ym = y[:, lo:]
                e = probe(f"  + GELU(tanh) over the mlp columns of {name}", lambda: torch.nn.functional.gelu(ym, approximate="tanh"), 0, 1.5)
                extra.append(dict(what="torch GELU(tanh) over the activation columns (a copy out of the strided view)", ms=e["ms_per_launch"]))
            if epi == ops.EPI_BIAS_GATE_RES:
                e = probe(f"  + gated residual of {name}", lambda: ops.gate_residual(y.view(1, m, N), gate, res.view(1, m, N), out=out.view(1, m, N)), 0, 1.5)
                extra.append(dict(what="tfx_gate_residual (res + gate * y)", ms=e["ms_per_launch"]))
            r2["extra_passes"] = extra
            r2["fair_ms_per_launch"] = round(r2["ms_per_launch"] + sum(e["ms"] for e in extra), 4)
            r2["fair_tflops"] = round(fl / (r2["fair_ms_per_launch"] * 1e-3) / 1e12, 1)
            rows.append(r2)
            del y
        del x, w, out, res
    own = [r for r in rows if r["tag"] == a.tag]
    tot = sum(r["share"] for r in own)
    summary = dict(tag=a.tag, flop_weighted_tflops=round(tot / sum(r["share"] / r["tflops"] for r in own), 1))
    hb = [r for r in rows if r["tag"] == "hipblaslt"]
    if hb:
        summary["hipblaslt_bias_only_flop_weighted_tflops"] = round(tot / sum(r["share"] / r["tflops"] for r in hb), 1)
        summary["hipblaslt_fair_flop_weighted_tflops"] = round(tot / sum(r["share"] / r["fair_tflops"] for r in hb), 1)
        summary["ratio_vs_bias_only"] = round(summary["flop_weighted_tflops"] / summary["hipblaslt_bias_only_flop_weighted_tflops"], 4)
        summary["ratio_vs_fair"] = round(summary["flop_weighted_tflops"] / summary["hipblaslt_fair_flop_weighted_tflops"], 4)
    print(json.dumps(summary), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
        f.write(json.dumps(summary) + "\n")


if __name__ == "__main__":
    main()
