#!/usr/bin/env python3
"""Sustained (power-capped steady state) rate of the persistent GEMM on every GEMM shape of the P1024 / batch-8 DiT step, with
the epilogue each shape carries in the model; rocm-smi sampled alongside.  Run once per library build to A/B kernels:

    TFX_LIB=textflux_amd/libtextflux_hip_mfma32.so python tools/gemm_shapes_power.py --tag mfma32x32x16
    python tools/gemm_shapes_power.py --tag mfma16x16x32"""
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
    ap.add_argument("--out", default="gpurun_out/r03_gemm_shapes.jsonl")
    ap.add_argument("--hipblaslt", action="store_true")
    ap.add_argument("--place", type=int, default=0, help="tfx_set_option gemm_place (bench knob)")
    ap.add_argument("--opt", action="append", default=[], help="name=value for tfx_set_option (bench knobs)")
    a = ap.parse_args()
    for kv in a.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if a.place:
        ops.set_option("gemm_place", a.place)
    M = 36864          # 8 x 4608 token rows
    # (name, rows, N, K, epilogue, FLOP share of a P1024 forward in units of D^2 per token)
    shapes = [("double qkv (img)", 32768, 3 * D, D, ops.EPI_BIAS, 19 * 3), ("double ff1 (img)", 32768, 4 * D, D, ops.EPI_BIAS_GELU, 19 * 4),
              ("double ff2 (img)", 32768, D, 4 * D, ops.EPI_BIAS_GATE_RES, 19 * 4), ("double out (img)", 32768, D, D, ops.EPI_BIAS_GATE_RES, 19 * 1),
              ("single qkv|mlp", M, 7 * D, D, ops.EPI_BIAS_GELU, 38 * 7), ("single proj_out", M, D, 5 * D, ops.EPI_BIAS_GATE_RES, 38 * 5)]
    rows = []
    for name, m, N, K, epi, share in shapes:
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
        r = probe(f"{a.tag} {name} {m}x{N}x{K}", lambda: ops.gemm(x, w, b, out=out, **kw), fl, a.secs)
        r.update(tag=a.tag, shape=name, M=m, N=N, K=K, share=share)
        rows.append(r)
        if a.hipblaslt:
            r2 = probe(f"hipBLASLt {name} {m}x{N}x{K}", lambda: torch.nn.functional.linear(x, w, b), fl, a.secs)
            r2.update(tag="hipblaslt", shape=name, M=m, N=N, K=K, share=share)
            rows.append(r2)
        del x, w, out, res
    own = [r for r in rows if r["tag"] == a.tag]
    tot = sum(r["share"] for r in own)
    wavg = tot / sum(r["share"] / r["tflops"] for r in own)
    print(json.dumps(dict(tag=a.tag, flop_weighted_tflops=round(wavg, 1))), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "a") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
        f.write(json.dumps(dict(tag=a.tag, flop_weighted_tflops=round(wavg, 1))) + "\n")


if __name__ == "__main__":
    main()
