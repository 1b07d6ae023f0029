"""Where an attention item's time goes (bench library: `make -C textflux_amd/csrc bench`, TFX_LIB=textflux_amd/libtextflux_hip_bench.so):
s_memtime stamps of wave 0 -- prologue (item start .. first tile), tile loop, output (.. item end) -- summed over all items of a launch."""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops, _lib
lib = _lib.lib()
fn = lib.tfx_bench_attn_timers
fn.argtypes = [ctypes.c_void_p]; fn.restype = None
B, H, D = 8, 24, 3072
for zero in (True, False):
    for N in (1152, 4608):
        y = torch.zeros(B, N, 3 * D, device="cuda", dtype=torch.bfloat16) if zero else torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16)
        q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
        o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
        for pers in (1, 0):
            ops.set_option("attention_persistent", pers)
            for _ in range(3):
                ops.attention(q, k, v, out=o, score_bound=20.0)
            tim = torch.zeros(4, dtype=torch.int64, device="cuda")
            fn(tim.data_ptr())
            ops.attention(q, k, v, out=o, score_bound=20.0)
            torch.cuda.synchronize()
            fn(None)
            t = tim.tolist()
            items = max(t[3], 1)
            # s_memtime counts core clock cycles on this part (the same counts on zero and on power-capped random data)
            print(json.dumps(dict(data="zero" if zero else "random", N=N, persistent=pers, items=t[3], tiles=(N + 63) // 64,
                                  cycles_prologue=round(t[0] / items), cycles_tile_loop=round(t[1] / items),
                                  cycles_per_tile=round(t[1] / items / ((N + 63) // 64)), cycles_output=round(t[2] / items))), flush=True)
ops.set_option("attention_persistent", 1)
