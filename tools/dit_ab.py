"""In-process A/B of library options on the real DiT forward (B=8, 1024x1024): alternates configurations on ONE box and
process, so box-to-box variance (+-2 %) drops out.  usage: python tools/dit_ab.py name=v1,v2 [name=v1,v2 ...] [--fp8]"""
import sys, os, time, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
from textflux_amd.transformer import FluxTransformer2DModel
BF = torch.bfloat16
args = [a for a in sys.argv[1:] if not a.startswith("--")]
fp8 = "--fp8" in sys.argv
knobs = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in args]
dev = torch.device("cuda")
B, S, T = 8, 4096, 512
m = FluxTransformer2DModel(in_channels=384, out_channels=64, guidance_embeds=True).init_random_(seed=1, device=dev)
if fp8:
    m.enable_fp8()
ses = m.session(B, S, T)
g = torch.Generator().manual_seed(0)
pe = (torch.randn(B, T, 4096, generator=g) * 0.1).to(BF).to(dev)
ids_img = torch.zeros(S, 3); ids_img[:, 1] = torch.arange(S) // 64; ids_img[:, 2] = torch.arange(S) % 64
ses.set_conditioning(pe, torch.zeros(T, 3), ids_img)
ses.xin.copy_(torch.randn(B, S, 384, generator=g).to(BF))
t = torch.full((B,), 500.0, device=dev); gd = torch.full((B,), 29952.0, device=dev)
mod = m.modulation(m.temb(t, gd, torch.randn(B, 768, generator=g).to(BF).to(dev)))
combos = list(itertools.product(*[v for _, v in knobs])) or [()]
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        ses.run(mod)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
run(2)
res = {c: [] for c in combos}
for rnd in range(4):
    for c in combos:
        for (name, _), v in zip(knobs, c):
            ops.set_option(name, v)
        run(1)
        res[c].append(run(4))
for c in combos:
    ts = sorted(res[c])
    print(dict(zip([k for k, _ in knobs], c)), "ms/forward: median", round(1e3 * ts[len(ts) // 2], 2), "min", round(1e3 * ts[0], 2), flush=True)
