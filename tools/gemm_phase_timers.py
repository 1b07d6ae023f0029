#!/usr/bin/env python3
"""Where a tile's time goes in the persistent GEMM: s_memtime sums of wave 0 of each row group (bench library only) for the
K loop, the epilogue's drain of outstanding requests, and the epilogue's convert / stage / store part, per shape and epilogue.

    make -C textflux_amd/csrc bench && TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/gemm_phase_timers.py"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16
D = 3072


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/gemm_phase_timers.json")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--place", type=int, default=0, help="tfx_set_option gemm_place (bench knob)")
    ap.add_argument("--opt", action="append", default=[], help="name=value for tfx_set_option (bench knobs)")
    a = ap.parse_args()
    for kv in a.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if a.place:
        ops.set_option("gemm_place", a.place)
    lib = _lib.lib()
    fn = lib.tfx_bench_gemm_timers
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = None
    tim = torch.zeros(8, dtype=torch.int64, device="cuda")
    M = 32768
    cases = [("K3072 N9216 bias", M, 3 * D, D, ops.EPI_BIAS), ("K3072 N12288 bias", M, 4 * D, D, ops.EPI_BIAS),
             ("K3072 N12288 gelu", M, 4 * D, D, ops.EPI_BIAS_GELU), ("K3072 N3072 gate_res", M, D, D, ops.EPI_BIAS_GATE_RES),
             ("K12288 N3072 gate_res", M, D, 4 * D, ops.EPI_BIAS_GATE_RES), ("K12288 N3072 bias", M, D, 4 * D, ops.EPI_BIAS),
             ("K3072 N9216 qkn [k|v|q]", M, 3 * D, D, "qkn"), ("K3072 N21504 qkn+gelu [k|v|q|mlp]", 36864, 7 * D, D, "qkn_gelu")]
    qfn = lib.tfx_bench_gemm_qkn
    qfn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    qfn.restype = None
    nw = torch.ones(128, dtype=BF, device="cuda")
    cs = torch.randn(36864, 64, 2, device="cuda")
    rows = []
    for name, m, N, K, epi in cases:
        x = torch.randn(m, K, device="cuda").to(BF)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
        b = torch.randn(N, device="cuda").to(BF)
        out = torch.empty(m, N, dtype=BF, device="cuda")
        qkn = isinstance(epi, str)
        kw = dict(epilogue=(ops.EPI_BIAS_GELU if qkn else epi))
        if qkn:
            kw["gelu_from_col"] = 3 * D if epi == "qkn_gelu" else N
            qfn(nw.data_ptr(), nw.data_ptr(), cs.data_ptr(), D)
        if epi == ops.EPI_BIAS_GATE_RES:
            kw.update(gate=torch.randn(1, N, device="cuda").to(BF), res=torch.randn(m, N, device="cuda").to(BF))
        fn(None)
        for _ in range(5):
            ops.gemm(x, w, b, out=out, **kw)
        torch.cuda.synchronize()
        tim.zero_()
        fn(tim.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.gemm(x, w, b, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        fn(None)
        qfn(None, None, None, 0)
        t = tim.tolist()
        ms = e0.elapsed_time(e1) / a.reps
        r = dict(case=name, ms=round(ms, 4), tflops=round(2.0 * m * N * K / ms / 1e9, 1))
        for g in (0, 1):
            kl, dr, ep, n = t[g * 4:g * 4 + 4]
            tot = kl + dr + ep
            r[f"g{g}"] = dict(tiles=n, ticks_per_tile=round(tot / max(n, 1), 1), k_loop=round(kl / tot, 4), drain=round(dr / tot, 4),
                              epilogue=round(ep / tot, 4), k_tiles_equiv_of_overhead=round((dr + ep) / (kl / (K // 64)), 2))
        print(json.dumps(r), flush=True)
        rows.append(r)
        del x, w, out
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
