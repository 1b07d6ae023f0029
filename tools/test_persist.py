import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
torch.manual_seed(0)
ok = True
for (B, M, N, K) in [(1, 36864, 3072, 256), (8, 4608, 3072, 3072), (3, 5120, 9216, 384), (1, 8192, 12288, 128), (2, 5248, 3072, 384), (1, 1152, 9216, 256), (2, 1000, 3136, 128), (1, 512, 3072, 3072), (1, 8, 64, 128), (3, 4736, 3328, 256)]:
    a = torch.randn(B, M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda").to(BF); gate = torch.randn(B, N, device="cuda").to(BF)
    res = torch.randn(B, M, N, device="cuda").to(BF)
    for epi, kw in [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, min(3072, (N // 256 - 1) * 256)))), (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)),
                    (ops.EPI_BIAS_RES, dict(res=res))]:
        o2 = torch.full((B, M, N), 7.0, dtype=BF, device="cuda")
        ops.gemm(a, w, bias, out=o2, epilogue=epi, variant=2, **kw)
        for place in (1, 2):
            ops.set_option("gemm_place", place)
            o3 = torch.full((B, M, N), 9.0, dtype=BF, device="cuda")
            ops.gemm(a, w, bias, out=o3, epilogue=epi, variant=3, **kw)
            torch.cuda.synchronize()
            same = torch.equal(o2, o3)
            ok &= same
            print((B, M, N, K), "epi", epi, "place", place, "bit-identical" if same else f"MISMATCH maxdiff {(o2.float() - o3.float()).abs().max().item()} n={(o2 != o3).sum().item()}", flush=True)
print("ALL OK" if ok else "FAILED")
for (M, N, K) in [(36864, 9216, 3072), (36864, 3072, 12288), (36864, 21504, 3072), (4096, 9216, 3072), (512, 9216, 3072), (5248, 21504, 3072), (1664, 21504, 3072), (1664, 3072, 15360)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    for rep in range(2):
        for v, place in ((2, 0), (3, 1), (3, 2)):
            ops.set_option("gemm_place", place)
            t = timeit(lambda: ops.gemm(x, w, b, out=out, variant=v), iters=20)
            print(dict(M=M, N=N, K=K, variant=v, place=place, ms=round(t * 1e3, 4), tflops=round(2.0 * M * N * K / t / 1e12, 1)), flush=True)
