"""Does the row pitch of the GEMM operands matter (L2 / Infinity-Cache channel interleave)?  The persistent GEMM on two block shapes with the
activation and / or weight rows padded by a few 128-byte lines; sustained timing, random bf16 data.  usage: python tools/gemm_pitch_sweep.py [secs]"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for name, M, N, K, epi in [("qkv (double img)", 32768, 9216, 3072, ops.EPI_BIAS), ("qkv_mlp (single)", 36864, 21504, 3072, ops.EPI_BIAS_GELU),
                           ("ff2 (double img)", 32768, 3072, 12288, ops.EPI_BIAS)]:
    b = torch.randn(N, device="cuda").to(BF)
    row = {}
    for pa, pw, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (128, 128, 0), (192, 192, 0), (0, 0, 64), (64, 64, 64)]:
        xs = torch.randn(M, K + pa, device="cuda").to(BF); x = xs[:, :K]
        ws = (torch.randn(N, K + pw, device="cuda") * 0.02).to(BF); w = ws[:, :K]
        outs = torch.empty(M, N + pc, dtype=BF, device="cuda"); out = outs[:, :N]
        kw = dict(out=out, epilogue=epi, gelu_from_col=0)
        try:
            for _ in range(3):
                ops.gemm(x, w, b, **kw)
        except Exception as e:
            row[f"a+{pa} w+{pw} c+{pc}"] = str(e)[:60]; continue
        torch.cuda.synchronize()
        t0, n = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(20):
                ops.gemm(x, w, b, **kw)
            torch.cuda.synchronize()
            n += 20
        dt = (time.time() - t0) / n
        row[f"a+{pa} w+{pw} c+{pc}"] = round(2.0 * M * N * K / dt / 1e12, 1)
        del xs, ws, outs
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "tflops_by_row_padding_elems": row}), flush=True)
