"""VAE encode / decode at 1024 x 1024 (default batch 8) on the HIP path, with and without the pixel-pair form of the <= 128-channel
3 x 3 layers (AutoencoderKL.pair_convs).  usage: python tools/vae_bench.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd.vae import AutoencoderKL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(B, 3, 1024, 1024, device="cuda").to(torch.bfloat16) * 2 - 1
z = torch.randn(B, 16, 128, 128, device="cuda").to(torch.bfloat16)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n
outs = {}
for pair in (True, False):
    vae = AutoencoderKL()
    vae.pair_convs = pair
    vae.init_random_(seed=7, device="cuda")
    print(f"B={B} pair_convs={pair}: encode {t(lambda: vae.encode(x))*1e3:.1f} ms  decode {t(lambda: vae.decode(z))*1e3:.1f} ms", flush=True)
    outs[pair] = vae.decode(z[:1]).sample.float() if hasattr(vae.decode(z[:1]), "sample") else vae.decode(z[:1])[0].float()
d = (outs[True] - outs[False]).abs()
print(f"decode pair vs one-pixel form: max |d| {d.max().item():.3e}, mean {d.mean().item():.3e} (|out| mean {outs[False].abs().mean().item():.3e})")
