import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
torch.manual_seed(0)
ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
ok = True
for (B, M, N, K) in [(1, 1664, 3072, 3072), (1, 512, 9216, 3072), (1, 1152, 3072, 12288), (1, 1664, 3072, 15360), (2, 300, 3136, 4096), (1, 512, 12288, 3072)]:
    a = torch.randn(B, M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.03).to(BF)
    bias = torch.randn(N, device="cuda").to(BF); gate = torch.randn(B, N, device="cuda").to(BF); res = torch.randn(B, M, N, device="cuda").to(BF)
    for epi, kw in [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, (N // 256 - 2) * 256))), (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)),
                    (ops.EPI_BIAS_RES, dict(res=res))]:
        o1 = ops.gemm(a, w, bias, epilogue=epi, variant=3, **kw)                    # persistent, unsplit
        o2 = ops.gemm(a, w, bias, epilogue=epi, variant=1, workspace=ws, **kw)      # auto: split-K with workspace
        d = (o1.float() - o2.float()).abs()
        rel = d.max().item() / o1.float().abs().max().item()
        good = rel < 8e-3 and torch.isfinite(o2).all().item()
        ok &= good
        print((B, M, N, K), "epi", epi, "identical" if torch.equal(o1, o2) else f"max diff/max {rel:.2e} mean {d.mean().item():.2e}", "OK" if good else "BAD", flush=True)
    t1 = timeit(lambda: ops.gemm(a, w, bias, variant=3), iters=20)
    t2 = timeit(lambda: ops.gemm(a, w, bias, variant=1, workspace=ws), iters=20)
    print(f"   unsplit {t1*1e6:.1f} us ({2.0*B*M*N*K/t1/1e12:.0f} TF)  split {t2*1e6:.1f} us ({2.0*B*M*N*K/t2/1e12:.0f} TF)", flush=True)
print("ALL OK" if ok else "FAILED")
