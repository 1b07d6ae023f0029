#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the MI355X (TFLOP/s for GEMM / attention, GB/s for the HBM-bound kernels).
Usage: python tools/bench_kernels.py [--quick] ; appends JSON lines to gpurun_out/kbench.jsonl"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops

BF = torch.bfloat16
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kbench.jsonl")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def emit(rec):
    print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "a") as f:
        f.write(json.dumps(rec) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = "cuda"
    D = 3072
    rows = [4608] if a.quick else [4608, 4608 * 8]
    if a.only == 'attention':
        rows = []
    shapes = [(3 * D, D), (D, D), (4 * D, D), (D, 4 * D), (7 * D, D), (D, 5 * D)]
    for M in rows:
        for N, K in shapes:
            x = torch.randn(M, K, device=dev).to(BF)
            w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
            b = torch.randn(N, device=dev).to(BF)
            out = torch.empty(M, N, dtype=BF, device=dev)
            t = timeit(lambda: ops.gemm(x, w, b, out=out, variant=1))
            emit(dict(tag=a.tag, kernel="gemm8p", M=M, N=N, K=K, ms=t * 1e3, tflops=2.0 * M * N * K / t / 1e12))
            t2 = timeit(lambda: torch.nn.functional.linear(x, w, b), iters=5)
            emit(dict(tag=a.tag, kernel="torch.linear(hipblaslt)", M=M, N=N, K=K, ms=t2 * 1e3,
                      tflops=2.0 * M * N * K / t2 / 1e12))
            del x, w, b, out
    for B in ([1] if a.quick else [1, 8]):
        for N in (4608, 8704) if not a.quick else (4608,):
            H = 24
            y = torch.randn(B, N, 3 * D, device=dev).to(BF)
            o = torch.empty(B, N, D, dtype=BF, device=dev)
            q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
            fl = 4.0 * B * H * N * N * 128
            for nw in (9, 8):
                ops.set_option("attention_waves", nw)
                t = timeit(lambda: ops.attention(q, k, v, out=o))
                emit(dict(tag=a.tag, kernel=f"attention(nw={nw})", B=B, N=N, ms=t * 1e3, tflops=fl / t / 1e12))
            ops.set_option("attention_waves", 8)
            for abl, nm in ((1, "no-softmax"), (2, "K-frags-once"), (4, "V-frags-once"), (8, "no-staging"), (15, "mfma-only")):
                ops.set_option("attention_ablation", abl)
                t = timeit(lambda: ops.attention(q, k, v, out=o))
                emit(dict(tag=a.tag, kernel=f"attention[{nm}]", B=B, N=N, ms=t * 1e3, tflops=fl / t / 1e12))
            ops.set_option("attention_ablation", 0)
            ops.set_option("attention_waves", 8)
            qh, kh, vh = (z.reshape(B, N, H, 128).transpose(1, 2) for z in (q, k, v))
            t2 = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh), iters=5)
            emit(dict(tag=a.tag, kernel="torch.sdpa", B=B, N=N, ms=t2 * 1e3, tflops=fl / t2 / 1e12))
            del y, o
    B, N = 8, 4608
    x = torch.randn(B, N, D, device=dev).to(BF)
    sh, sc = torch.randn(B, D, device=dev).to(BF), torch.randn(B, D, device=dev).to(BF)
    o = torch.empty_like(x)
    t = timeit(lambda: ops.ln_modulate(x, sh, sc, out=o))
    emit(dict(tag=a.tag, kernel="ln_modulate", rows=B * N, ms=t * 1e3, gbs=2.0 * x.numel() * 2 / t / 1e9))
    y = torch.randn(B, N, 7 * D, device=dev).to(BF)
    wn = torch.ones(128, device=dev).to(BF)
    cos, sin = torch.randn(N, 128, device=dev), torch.randn(N, 128, device=dev)
    t = timeit(lambda: ops.rmsnorm_rope_(y, 2 * D, 0, 24, 512, wn, wn, wn, wn, cos, sin))
    emit(dict(tag=a.tag, kernel="rmsnorm_rope", rows=B * N, ms=t * 1e3, gbs=2.0 * (B * N * 2 * D) * 2 / t / 1e9))


if __name__ == "__main__":
    main()
