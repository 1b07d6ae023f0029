"""Launch time of the head-dim-128 attention with whole (b, h, q-tile) items vs the stream-K dealing (round 6, option attention_streamk), on
random data, alternating, HIP events over 20 launches per arm and 3 rounds.  usage: python tools/attn_streamk_time.py [B H N ...]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
shapes = [(8, 24, 4608), (1, 24, 4608), (1, 24, 1664), (1, 24, 5248), (1, 24, 8704), (8, 24, 8704), (4, 24, 5248), (2, 24, 4608), (1, 24, 3200)]
if len(sys.argv) > 3:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
ws = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
for (B, H, N) in shapes:
    D = H * 128
    y = (torch.randn(B, N, 3 * D, device="cuda") * 1.0).to(BF)
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
    out = torch.empty(B, N, D, device="cuda", dtype=BF)
    res = {0: [], 1: [], 2: []}
    for rnd in range(3):
        for lvl in (0, 1, 2):
            ops.set_option("attention_streamk", lvl)
            ops.attention(q, k, v, out=out, score_bound=30.0, workspace=ws)
            st, en = torch.cuda.Event(True), torch.cuda.Event(True)
            st.record()
            for _ in range(20):
                ops.attention(q, k, v, out=out, score_bound=30.0, workspace=ws)
            en.record(); torch.cuda.synchronize()
            res[lvl].append(st.elapsed_time(en) / 20)
    fl = 4.0 * N * N * 128 * H * B
    med = {l: sorted(v)[1] for l, v in res.items()}
    print(f"B {B} H {H} N {N}: whole items {med[0]*1e3:.1f} us ({fl/med[0]/1e9:.0f} TFLOP/s) | auto {med[1]*1e3:.1f} us | dealt (forced) {med[2]*1e3:.1f} us ({fl/med[2]/1e9:.0f} TFLOP/s): {100*(med[2]/med[0]-1):+.1f} %", flush=True)
ops.set_option("attention_streamk", 1)
