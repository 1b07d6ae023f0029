import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops, _lib as L
BF = torch.bfloat16
B, N, H, D = 8, 4608, 24, 3072
y = torch.randn(B, N, 3 * D, device="cuda").to(BF)
o = torch.empty(B, N, D, dtype=BF, device="cuda")
q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
nblk = B * H * ((N + 255) // 256)
dbg = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.tfx_debug_attention_timing.argtypes = [C.c_void_p]
lib.tfx_debug_attention_timing(dbg.data_ptr())
ops.set_option('attention_waves', 16)
for _ in range(3):
    ops.attention(q, k, v, out=o)
torch.cuda.synchronize()
d = dbg.view(nblk, 8, 4).double()
nkv = (N + 63) // 64
per = d / nkv
for gi, name in ((slice(0, 4), "G0"), (slice(4, 8), "G1")):
    print(name, "ticks per tile: PA %.0f  barrier1 %.0f  PB %.0f  barrier2 %.0f   total %.0f" % tuple(
        [per[:, gi, i].mean().item() for i in range(4)] + [per[:, gi].sum(-1).mean().item()]))
lib.tfx_debug_attention_timing(None)
# lock-step kernel, section timing (ablation bit 4)
dbg.zero_()
lib.tfx_debug_attention_timing(dbg.data_ptr())
ops.set_option("attention_waves", 8)
ops.set_option("attention_ablation", 16)
for _ in range(3):
    ops.attention(q, k, v, out=o)
torch.cuda.synchronize()
per = dbg.view(nblk, 8, 4).double() / nkv
print("lock-step ticks per tile: QK %.0f  softmax %.0f  PV %.0f  staging+barrier %.0f  total %.0f" % tuple(
    [per[:, :, i].mean().item() for i in range(4)] + [per.sum(-1).mean().item()]))
ops.set_option("attention_ablation", 0)
lib.tfx_debug_attention_timing(None)
