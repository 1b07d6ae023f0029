#!/usr/bin/env python3
"""Round 6, VERDICT round 5 item 1(a), attention side: sustained 3 s loops (rocm-smi sampled, power-capped steady state) of the default
attention kernel at the headline shape (B = 8, H = 24, N = 4608, reference-free stream) from the product library and from the timing-ablation
build libtextflux_hip_exp_abl1024.so (-DW4_ABL=1024: the Q-side per-head RMSNorm + RoPE executed in the kernel's Q prologue, table loads
omitted = a lower bound on its price), alternating, each library in its own process.

    tools/build_variant.sh abl1024 "-DW4_ABL=1024" && python tools/attn_q_prologue_ab.py gpurun_out/r06_attn_q_prologue.json"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, REPO)
    from textflux_amd import ops
    from tools.power_profile import probe
    B, N, D = 8, 4608, 3072
    y = torch.randn(B, N, 3 * D, device="cuda").to(torch.bfloat16)
    q, k, v = y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D]
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    r = probe(os.path.basename(os.environ.get("TFX_LIB", "default")), lambda: ops.attention(q, k, v, out=o, score_bound=20.0),
              4.0 * B * 24 * N * N * 128, 3.0)
    print("RESULT " + json.dumps(r), flush=True)
else:
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "r06_attn_q_prologue.json")
    rows = []
    for rnd in range(2):
        for lib in ("", os.path.join(REPO, "textflux_amd", "libtextflux_hip_exp_abl1024.so")):
            env = dict(os.environ)
            env.pop("TFX_LIB", None)
            if lib:
                env["TFX_LIB"] = lib
            p = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            for line in p.stdout.splitlines():
                if line.startswith("RESULT "):
                    r = json.loads(line[7:])
                    r["round"] = rnd
                    rows.append(r)
                    print(r, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)
