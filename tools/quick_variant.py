import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
M, N, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
shapes = {"a": (36864, 9216, 3072), "b": (36864, 3072, 12288), "c": (36864, 21504, 3072), "d": (4608, 9216, 3072)}
for v in [int(a) for a in sys.argv[1:] if not a.startswith("gm")]:
    pass
    for _ in range(3):
        ops.gemm(x, w, b, out=out, variant=v)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        ops.gemm(x, w, b, out=out, variant=v)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 20
    print(f"variant {v}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TF", flush=True)

for a in sys.argv[1:]:
    if a.startswith("gm"):
        for name, (M, N, K) in shapes.items():
            x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
            b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
            res = []
            for gm in (1, 2, 4, 8, 16, 32, 8):
                ops.set_option("gemm_group_m", gm)
                for _ in range(3): ops.gemm(x, w, b, out=out, variant=1)
                torch.cuda.synchronize(); t0 = time.time()
                for _ in range(20): ops.gemm(x, w, b, out=out, variant=1)
                torch.cuda.synchronize(); dt = (time.time() - t0) / 20
                res.append((gm, round(2.0 * M * N * K / dt / 1e12)))
            print(name, (M, N, K), res, flush=True)
