"""Experiment (round 4): does running the batch as TWO half-batch forwards on two streams fill the partly empty last rounds of the
persistent GEMMs / attention (2.4 % / 3.7 % of their time at batch 8) with the other half's work?  One B = 8 session against two
B = 4 sessions issued on two streams, eager launches, same box and process.  usage: python tools/dit_two_streams.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd.transformer import FluxTransformer2DModel, DitSession
BF = torch.bfloat16
dev = torch.device("cuda")
S, T = 4096, 512
m = FluxTransformer2DModel(in_channels=384, out_channels=64, guidance_embeds=True).init_random_(seed=1, device=dev)
g = torch.Generator().manual_seed(0)
def make(B):
    ses = DitSession(m, B, S, T)
    pe = (torch.randn(B, T, 4096, generator=g) * 0.1).to(BF).to(dev)
    ids_img = torch.zeros(S, 3); ids_img[:, 1] = torch.arange(S) // 64; ids_img[:, 2] = torch.arange(S) % 64
    ses.set_conditioning(pe, torch.zeros(T, 3), ids_img)
    ses.xin.copy_(torch.randn(B, S, 384, generator=g).to(BF))
    t = torch.full((B,), 500.0, device=dev); gd = torch.full((B,), 29952.0, device=dev)
    mod = m.modulation(m.temb(t, gd, torch.randn(B, 768, generator=g).to(BF).to(dev)))
    return ses, mod
full, mod8 = make(8)
h1, mod4a = make(4)
h2, mod4b = make(4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_full(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        full.run(mod8)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
def run_halves(n, offset=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1):
            h1.run(mod4a)
        with torch.cuda.stream(s2):
            h2.run(mod4b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
def run_halves_serial(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        h1.run(mod4a); h2.run(mod4b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
run_full(2); run_halves(2); run_halves_serial(1)
for rnd in range(3):
    print("B=8 one stream: %.2f ms | 2 x B=4 on two streams: %.2f ms | 2 x B=4 one stream: %.2f ms" % (1e3 * run_full(4), 1e3 * run_halves(4), 1e3 * run_halves_serial(4)), flush=True)
