"""GroupNorm(+SiLU) launch times at the VAE's shapes (batch 8, 1024 x 1024 image): HIP events around 10 launches each, and the HBM rate
2 x B x HW x C x 2 bytes (apply) / 1 x (statistics) would need.  usage: python tools/gn_time.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
import hashlib
torch.manual_seed(0)
for (B, HW, C) in ((8, 1024 * 1024, 128), (8, 1024 * 1024, 256), (8, 512 * 512, 256), (8, 512 * 512, 512), (8, 256 * 256, 512), (8, 128 * 128, 512)):
    x = torch.randn(B, HW, C, device="cuda", dtype=BF, generator=torch.Generator(device='cuda').manual_seed(HW + C))
    ga = torch.randn(C, device="cuda", dtype=BF); be = torch.randn(C, device="cuda", dtype=BF)
    out = torch.empty_like(x)
    ops.groupnorm_nhwc(x, ga, be, 32, True, out=out)
    st, en = torch.cuda.Event(True), torch.cuda.Event(True)
    st.record()
    for _ in range(10):
        ops.groupnorm_nhwc(x, ga, be, 32, True, out=out)
    en.record(); torch.cuda.synchronize()
    ms = st.elapsed_time(en) / 10
    gb = 3 * B * HW * C * 2 / 1e9
    out2 = torch.empty_like(x)
    ops.groupnorm_nhwc(x, ga, be, 32, True, out=out2)
    h = hashlib.sha1(out[:2].cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    hx = hashlib.sha1(x[:1].cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:8]
    print(f"B {B} HW {HW} C {C}: rerun identical {torch.equal(out, out2)}, sha1(out[:2]) {h}, sha1(x[:1]) {hx}", flush=True)
    print(f"B {B} HW {HW} C {C}: {ms*1e3:.0f} us per GroupNorm+SiLU (3 passes over {B*HW*C*2/1e9:.2f} GB: {gb/ms:.2f} TB/s)", flush=True)
