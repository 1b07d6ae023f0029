import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
F8 = torch.float8_e4m3fn
torch.manual_seed(0)
ok = True
for (B, M, N, K) in [(1, 512, 512, 256), (2, 1000, 3136, 512), (1, 4608, 9216, 3072), (2, 4736, 3072, 15360), (3, 300, 264, 12288)]:
    a = (torch.randn(B, M, K, device="cuda") * torch.rand(B, M, 1, device="cuda") * 3).to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda").to(BF); gate = torch.randn(B, N, device="cuda").to(BF); res = torch.randn(B, M, N, device="cuda").to(BF)
    aq, sa = ops.quantize_rows_fp8(a)
    wq, sw = ops.quantize_rows_fp8(w)
    # quantiser vs torch
    sa_ref = a.float().abs().amax(-1) / 448.0
    sa_ref = torch.where(sa_ref > 0, sa_ref, torch.ones_like(sa_ref))
    q_ref = (a.float() / sa_ref[..., None]).clamp(-448, 448).to(F8)
    same_scale = torch.allclose(sa, sa_ref, rtol=1e-6, atol=0)
    dq = aq.view(F8).float(); dr = q_ref.float()
    nmis = (dq != dr).sum().item()
    print((B, M, N, K), "quant: scale ok", same_scale, "byte mismatches", nmis, "of", dq.numel(), "max rel", ((dq - dr).abs() / dr.abs().clamp_min(1e-3)).max().item(), flush=True)
    ok &= same_scale and nmis <= dq.numel() * 1e-3
    lin = (dq @ wq.view(F8).float().T) * sa[..., None] * sw[None, None, :] + bias.float()
    for epi, kw, ref in [(ops.EPI_BIAS, {}, lin.to(BF)),
                         (ops.EPI_BIAS_GELU, dict(gelu_from_col=0), torch.nn.functional.gelu(lin, approximate="tanh").to(BF)),
                         (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res), (res.float() + (gate.float()[:, None] * lin.to(BF).float()).to(BF).float()).to(BF)),
                         (ops.EPI_BIAS_RES, dict(res=res), (res.float() + lin.to(BF).float()).to(BF))]:
        got = ops.gemm_fp8(aq, sa, wq, sw, bias, epilogue=epi, **kw)
        torch.cuda.synchronize()
        err = (got.float() - ref.float()).abs()
        rel = err.max().item() / ref.float().abs().max().item()
        good = rel < 1e-2 and torch.isfinite(got).all().item()
        ok &= good
        print("   epi", epi, "max err / max|ref|", f"{rel:.2e}", "mean err", f"{err.mean().item():.2e}", "OK" if good else "BAD", flush=True)
    # fp8 quantisation error against the bf16 GEMM (information only)
    full = a.float() @ w.float().T + bias.float()
    print("   fp8 vs bf16-operand GEMM: rel RMS error", f"{((lin - full).pow(2).mean().sqrt() / full.pow(2).mean().sqrt()).item():.3e}")
print("ALL OK" if ok else "FAILED")
for (M, N, K) in [(36864, 9216, 3072), (36864, 3072, 12288), (36864, 21504, 3072), (36864, 3072, 15360)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    xq, sx = ops.quantize_rows_fp8(x); wq, sw = ops.quantize_rows_fp8(w)
    for rep in range(2):
        t = timeit(lambda: ops.gemm_fp8(xq, sx, wq, sw, b, out=out), iters=20)
        t2 = timeit(lambda: ops.gemm(x, w, b, out=out), iters=20)
        t3 = timeit(lambda: ops.quantize_rows_fp8(x, out=xq, scale=sx), iters=20)
        print(dict(M=M, N=N, K=K, fp8_ms=round(t * 1e3, 4), fp8_tflops=round(2.0 * M * N * K / t / 1e12, 1), bf16_ms=round(t2 * 1e3, 4),
                   bf16_tflops=round(2.0 * M * N * K / t2 / 1e12, 1), quant_ms=round(t3 * 1e3, 4), quant_GBps=round(M * K * 3 / t3 / 1e9, 1)), flush=True)
