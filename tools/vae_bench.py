import os, sys, time, torch
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd.vae import AutoencoderKL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vae = AutoencoderKL().init_random_(seed=7, device="cuda")
x = torch.rand(B, 3, 1024, 1024, device="cuda").to(torch.bfloat16) * 2 - 1
z = torch.randn(B, 16, 128, 128, device="cuda").to(torch.bfloat16)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n
print(f"B={B} encode {t(lambda: vae.encode(x))*1e3:.1f} ms  decode {t(lambda: vae.decode(z))*1e3:.1f} ms", flush=True)
vae.use_hip = False
print(f"torch/MIOpen NCHW path: encode {t(lambda: vae.encode(x))*1e3:.1f} ms  decode {t(lambda: vae.decode(z))*1e3:.1f} ms", flush=True)
