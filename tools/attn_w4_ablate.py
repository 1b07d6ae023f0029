"""Timing ablations of attn_w4_kernel (results are WRONG by construction): one library per -DW4_ABL value, built by hand
(textflux_amd/libtextflux_hip_exp_abl<N>.so), timed in separate processes.  usage: python tools/attn_w4_ablate.py [mode ...]"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, REPO)
    from textflux_amd import ops
    from tools.bench_kernels import timeit
    B, N, D = 8, 4608, 3072
    zero = os.environ.get("W4_ZERO") == "1"
    y = torch.zeros(B, N, 3 * D, device="cuda", dtype=torch.bfloat16) if zero else torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    for mode in [int(a) for a in sys.argv[2:]]:
        ops.set_option("attention_waves", mode)
        sb = 20.0 if mode == 34 else 0.0        # 34 = the reference-free stream: needs the caller's score bound
        timeit(lambda: ops.attention(q, k, v, out=o, score_bound=sb), iters=10)
        t = min(timeit(lambda: ops.attention(q, k, v, out=o, score_bound=sb), iters=20) for _ in range(3))
        print(json.dumps(dict(lib=os.path.basename(os.environ.get("TFX_LIB", "default")), mode=mode, data="zero" if zero else "random", ms=round(t * 1e3, 4))), flush=True)
else:
    modes = sys.argv[1:] or ["30", "32"]
    abls = os.environ["W4_ABL_LIST"].split(",") if os.environ.get("W4_ABL_LIST") else ["1", "2", "4", "64", "320", "128", "8", "16", "6", "31", "512"]
    for abl in ([""] if os.environ.get("W4_NO_ABL") else [""] + abls):
        env = dict(os.environ)
        if abl:
            env["TFX_LIB"] = os.path.join(REPO, "textflux_amd", f"libtextflux_hip_exp_abl{abl}.so")
            if not os.path.exists(env["TFX_LIB"]):
                continue
        subprocess.run([sys.executable, __file__, "--child", *modes], env=env)
