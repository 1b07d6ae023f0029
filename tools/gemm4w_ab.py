"""A/B of the two persistent GEMM kernels (tfx_set_option gemm_waves 8 = ping-pong, 4 = one wave per SIMD) on the six GEMM shapes of
a P1024 / batch-8 step with the epilogue each carries in the model: bit-identity first, then alternating timed loops."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from tools.bench_kernels import timeit
BF = torch.bfloat16
D = 3072
shapes = [("double qkv (img)", 32768, 9216, 3072, ops.EPI_BIAS), ("double ff1", 32768, 12288, 3072, ops.EPI_BIAS_GELU),
          ("double ff2", 32768, 3072, 12288, ops.EPI_BIAS_GATE_RES), ("double out", 32768, 3072, 3072, ops.EPI_BIAS_GATE_RES),
          ("single qkv|mlp", 36864, 21504, 3072, ops.EPI_BIAS_GELU), ("single proj_out", 36864, 3072, 15360, ops.EPI_BIAS_GATE_RES)]
g = torch.Generator().manual_seed(0)
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, generator=g).to(BF).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(BF).cuda()
    bias = torch.randn(N, generator=g).to(BF).cuda()
    kw = {}
    if epi == ops.EPI_BIAS_GATE_RES:
        kw = dict(gate=torch.randn(1, N, generator=g).to(BF).cuda(), res=torch.randn(M, N, generator=g).to(BF).cuda())
    if epi == ops.EPI_BIAS_GELU:
        kw = dict(gelu_from_col=3 * D if N == 7 * D else 0)
    outs, ts = {}, {8: [], 4: []}
    for nw in (8, 4):
        ops.set_option("gemm_waves", nw)
        outs[nw] = ops.gemm(a.view(1, M, K), w, bias, epilogue=epi, variant=3, **({k: (v.view(1, M, N) if k == "res" else v) for k, v in kw.items()}))
    same = torch.equal(outs[8], outs[4])
    o = torch.empty(1, M, N, dtype=BF, device="cuda")
    kw2 = {k: (v.view(1, M, N) if k == "res" else v) for k, v in kw.items()}
    for rnd in range(3):
        for nw in (8, 4):
            ops.set_option("gemm_waves", nw)
            ts[nw].append(timeit(lambda: ops.gemm(a.view(1, M, K), w, bias, out=o, epilogue=epi, variant=3, **kw2), iters=20))
    fl = 2.0 * M * N * K
    print(json.dumps(dict(shape=name, M=M, N=N, K=K, bit_identical=same, tflops_8w=round(fl / min(ts[8]) / 1e12, 1), tflops_4w=round(fl / min(ts[4]) / 1e12, 1),
                          max_abs_diff=(outs[8].float() - outs[4].float()).abs().max().item())), flush=True)
    del a, w, o, outs
ops.set_option("gemm_waves", 8)
