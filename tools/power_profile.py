#!/usr/bin/env python3
"""Power evidence for the roofline discussion (DESIGN.md "Where the GEMM stands"): sustained loops of each kernel with
rocm-smi sampled alongside -> ms / launch, sclk, board W, J / launch, TFLOP/s, on random and on all-zero data.

    make -C textflux_amd/csrc bench          (the -DTFX_BENCH library carries the MFMA-only ablation)
    TFX_LIB=textflux_amd/libtextflux_hip_bench.so python tools/power_profile.py --out gpurun_out/r03_power.json

Rows: near-idle (a trivial kernel in a loop), pure MFMA (the one-tile GEMM with requests, LDS reads and barriers compiled
out: the ceiling the power cap leaves to ANY bf16 MFMA kernel on this data), the persistent GEMM (the product kernel),
hipBLASLt (torch.nn.functional.linear) on the same operands, the default attention kernel; the GEMM / attention rows
again on zero operands (no data-dependent switching power: what the kernels do when the cap does not bind).
`power_capped_peak_tflops` = the pure-MFMA row's rate on random data; bench.py reports it beside the nominal 2500."""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops  # noqa: E402

BF = torch.bfloat16


def _sample(stop, acc):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", r)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r)
            if sclk and pw:
                acc.append((int(sclk[0]), float(pw[0])))
        except Exception:
            pass
        time.sleep(0.25)


def probe(name, fn, flops, secs):
    stop, acc = threading.Event(), []
    th = threading.Thread(target=_sample, args=(stop, acc), daemon=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0, n = time.time(), 0
    try:
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
    except Exception as e:
        stop.set()
        print(name, "FAILED", e, flush=True)
        return dict(name=name, error=str(e))
    dt = time.time() - t0
    stop.set()
    th.join(timeout=3)
    a = acc[2:] or acc or [(-1, -1.0)]
    ms = dt / n * 1e3
    w = sum(x[1] for x in a) / len(a)
    rec = dict(name=name, ms_per_launch=round(ms, 4), sclk_mhz=round(sum(x[0] for x in a) / len(a)), board_w=round(w, 1),
               j_per_launch=round(ms * w / 1e3, 4), tflops=round(flops / (ms * 1e-3) / 1e12, 1) if flops else None,
               launches=n, power_samples=len(a))
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r03_power.json")
    ap.add_argument("--secs", type=float, default=3.0)
    a = ap.parse_args()
    M, N, K = 36864, 9216, 3072
    D = 3072
    gf = 2.0 * M * N * K
    af = 4.0 * 8 * 24 * 4608 * 4608 * 128
    rows = []
    tiny = torch.empty(1024, dtype=BF, device="cuda")
    rows.append(probe("near-idle (trivial kernel loop)", lambda: tiny.zero_(), 0, a.secs))
    have_abl = "bench" in os.path.basename(os.environ.get("TFX_LIB", ""))
    for data in ("random", "zero"):
        if data == "random":
            x = torch.randn(M, K, device="cuda").to(BF)
            w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
            y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF)
        else:
            x, w = torch.zeros(M, K, dtype=BF, device="cuda"), torch.zeros(N, K, dtype=BF, device="cuda")
            y = torch.zeros(8, 4608, 3 * D, dtype=BF, device="cuda")
        b = torch.randn(N, device="cuda").to(BF)
        out = torch.empty(M, N, dtype=BF, device="cuda")
        o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
        if have_abl:
            rows.append(probe(f"pure MFMA (one-tile GEMM, no requests / LDS reads / barriers), {data} data",
                              lambda: ops.gemm(x, w, b, out=out, variant=17), gf, a.secs))
        rows.append(probe(f"persistent GEMM {M}x{N}x{K}, {data} data", lambda: ops.gemm(x, w, b, out=out, variant=3), gf, a.secs))
        rows.append(probe(f"one-tile GEMM {M}x{N}x{K}, {data} data", lambda: ops.gemm(x, w, b, out=out, variant=2), gf, a.secs))
        rows.append(probe(f"hipBLASLt (torch.nn.functional.linear) {M}x{N}x{K}, {data} data",
                          lambda: torch.nn.functional.linear(x, w, b), gf, a.secs))
        rows.append(probe(f"attention (default kernel) B8 H24 N4608, {data} data",
                          lambda: ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o), af, a.secs))
        del x, w, y, out, o
    # round 6: the product library's MFMA-only kernel (tfx_mfma_peak_probe: the GEMM kernels' MFMA sections on register-resident operands, no
    # prologue, no epilogue, one workgroup per CU) next to the older "pure MFMA" row (the ONE-TILE GEMM with its requests, LDS reads and barriers
    # compiled out -- 5184 workgroups that still run their address arithmetic, prologue waits and a full epilogue each): bench.py's live
    # roofline.power_capped_peak is the former
    if hasattr(ops, "mfma_peak_probe"):
        import ctypes as C
        from textflux_amd import _lib as L
        g = torch.Generator().manual_seed(5)
        r32 = torch.randn(1 << 20, generator=g)
        for name, buf, f8 in (("bf16", r32.to(BF).cuda(), 0), ("e4m3", r32.to(torch.float8_e4m3fn).view(torch.uint8).cuda(), 1),
                              ("bf16, zero data", torch.zeros(1 << 20, dtype=BF, device="cuda"), 0)):
            fl = C.c_double()
            kt = 48 * 4096
            fn = lambda: L.check(L.lib().tfx_mfma_peak_probe(buf.data_ptr(), buf.numel() * buf.element_size(), f8, kt, C.byref(fl),
                                                             torch.cuda.current_stream().cuda_stream), "probe")
            fn(); torch.cuda.synchronize()
            rows.append(probe(f"tfx_mfma_peak_probe ({name}): MFMA sections only, registers only", fn, fl.value, a.secs))
    idle_w = rows[0].get("board_w")
    mf = next((r for r in rows if r["name"].startswith("pure MFMA") and "random" in r["name"] and "tflops" in r), None)
    rec = dict(device=torch.cuda.get_device_name(0), shape=dict(M=M, N=N, K=K), gemm_flops_per_launch=gf, attention_flops_per_launch=af,
               secs_per_row=a.secs, method="wall time of a sustained loop / launches; rocm-smi --showclocks --showpower sampled every 0.25 s "
                                           "during the loop (first two samples dropped); J = mean W x s",
               near_idle_w=idle_w, power_capped_peak_tflops=mf["tflops"] if mf else None,
               power_capped_peak_note="rate of the MFMA-only ablation on random bf16 operands: the one-tile GEMM without requests / LDS reads / barriers (still one workgroup per tile with its prologue and epilogue); the register-only stream is the tfx_mfma_peak_probe row",
               rows=rows)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
