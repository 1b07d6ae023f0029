"""In-kernel cycle accounting of the GEMM main loop (bench-only ablation bit 7) + tick-rate calibration."""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops, _lib as L
BF = torch.bfloat16
for (M, N, K) in [(36864, 9216, 3072)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device="cuda")
    g = L.GemmArgs()
    g.A, g.lda, g.a_bstride = x.data_ptr(), K, 0
    g.W, g.ldw, g.bias = w.data_ptr(), K, b.data_ptr()
    g.C, g.ldc, g.c_bstride = out.data_ptr(), N, 0
    g.M, g.N, g.K, g.batch = M, N, K, 1
    g.epilogue, g.gelu_from_col = 0, 0
    g.res, g.ldr, g.r_bstride = dbg.data_ptr(), 8, 0
    nt = K // 64
    for var, vname in ((138, "full"), (139, "no-prefetch"), (145, "mfma-only"), (154, "no-mfma")):
        for _ in range(3):
            L.check(L.lib().tfx_gemm_bf16(C.byref(g), var, ops._stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(L.lib().tfx_gemm_bf16(C.byref(g), var, ops._stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        d = dbg.view(nblk, 8, 4).double()
        tpb = d.sum(-1).mean().item()
        rounds = nblk / 256.0
        per = d / (nt * 4)
        print(f"[{vname}] kernel {ms:.3f} ms; loop ticks/block {tpb:.0f} ({tpb / nt:.0f} per K-tile; ideal 2048 MFMA cycles); "
              f"tick rate if blocks run back-to-back {tpb * rounds / (ms * 1e-3) / 1e9:.2f} GHz; "
              f"phase split wait/bar1/mfma/bar2 = {per[:, :, 0].mean():.0f}/{per[:, :, 1].mean():.0f}/{per[:, :, 2].mean():.0f}/{per[:, :, 3].mean():.0f}")
