#!/usr/bin/env python3
"""How far does the reference's bf16 run sit from ITSELF when only the fp32 summation order inside its matrix products changes?


The bf16-faithful CPU oracle (bit-exact restatement of the reference's bf16 run, tests/test_oracle_golden.py) is run a second time over the
full 19 + 38-block model and all 30 Euler steps of BASELINE config 2 (the g11 fixture's setting), with ONE change: every nn.Linear sums its
K products in a different order -- the K axis of the input and of the weight is permuted by the same fixed permutation, which leaves the
mathematical result, the kernel, the accumulator type (fp32) and every rounding point (one bf16 rounding per output element) untouched.
What differs afterwards is therefore exactly "fp32 summation order", the freedom ANY implementation of the reference's arithmetic has
(thread count, K blocking, split-K, MFMA tile order).  Two tables:

  free-running   both runs integrate their own trajectory from the same noise   -> the floor under the engine's free-running distance
  teacher-forced every step starts from the fixture's latents of the step before -> the floor under a per-forward comparison

    python tools/oracle_self_noise.py [--threads 8] [--steps 30]      # CPU only, ~15 min on 8 cores -> profiles/r05_oracle_self_noise.json
    python tools/oracle_self_noise.py --what all --modes teacher_forced --out profiles/r05_oracle_self_noise_all.json

--what all: the same freedom for every other reduction of the forward as well -- the KEY axis of scaled_dot_product_attention (k and v
permuted together: softmax(q k^T) v does not depend on the order of the keys), the feature axis of F.layer_norm and of the per-head RMSNorm
(permute, normalise, permute back).  Still the reference's own kernels, accumulator types and rounding points.

Reference: D/pipelines/flux/pipeline_flux_fill.py:2053-2112, D/models/transformers/transformer_flux.py:1028-1212.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.fulldepth_trajectory import FIXTURE, H, W, N_SCHED, inputs, seeded_weights   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--steps", type=int, default=N_SCHED)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r05_oracle_self_noise.json"))
    ap.add_argument("--what", choices=["linear", "all"], default="linear")
    ap.add_argument("--modes", nargs="+", default=["teacher_forced", "free_running"])
    a = ap.parse_args()
    from safetensors.torch import load_file
    from oracle import flux_oracle as fo
    from oracle import pipeline_oracle as po
    torch.set_num_threads(a.threads)
    ref = load_file(FIXTURE)["traj_bf16"]
    cfg, sd = seeded_weights()
    lat, mil, pe, pooled = inputs()

    # the permutation of every K axis (one per width), applied to the weights once and to the activations at every call
    perms = {}

    def perm(k):
        if k not in perms:
            perms[k] = torch.randperm(k, generator=torch.Generator().manual_seed(k))
        return perms[k]

    t0 = time.time()
    for name in list(sd):
        if name.endswith(".weight") and sd[name].dim() == 2:
            sd[name] = sd[name][:, perm(sd[name].shape[1])].contiguous()
    print(f"weights permuted {time.time() - t0:.0f} s", flush=True)
    plain = fo.linear

    def linear_permuted(x, sd_, name):
        return plain(x[..., perm(x.shape[-1])], sd_, name)

    fo.linear = linear_permuted
    if a.what == "all":
        import torch.nn.functional as F
        sdpa, rms, ln = F.scaled_dot_product_attention, fo.rms_norm, fo.layer_norm

        def inv(p):
            q = torch.empty_like(p)
            q[p] = torch.arange(p.numel())
            return q

        def sdpa_permuted(q, k, v, **kw):
            pk = perm(k.shape[2])
            return sdpa(q, k[:, :, pk], v[:, :, pk], **kw)

        def rms_permuted(x, weight, eps=1e-6):
            pf = perm(x.shape[-1])
            return rms(x[..., pf], weight[pf], eps)[..., inv(pf)]

        def ln_permuted(x, eps=1e-6):
            pf = perm(x.shape[-1])
            return ln(x[..., pf], eps)[..., inv(pf)]

        F.scaled_dot_product_attention, fo.rms_norm, fo.layer_norm = sdpa_permuted, rms_permuted, ln_permuted
    mae = lambda x, y: (x.float() - y.float()).abs().mean().item()
    rec = {"what": "bf16-faithful oracle vs ITSELF with the K axis of every nn.Linear permuted (same kernel, same fp32 accumulator, same single "
                   "bf16 rounding per output; only the fp32 summation order differs)"
                   + (", AND the key axis of every scaled_dot_product_attention, the feature axis of every LayerNorm / per-head RMSNorm permuted"
                      if a.what == "all" else "")
                   + ": full 19+38-block model, SL512 576x512 batch 1, the g11 fixture's weights / inputs / 30-step Euler schedule; latent MAE per step",
           "permuted": a.what,
           "threads": a.threads, "steps": a.steps, "north_star_tolerance": 1e-3, "tool": "tools/oracle_self_noise.py"}
    with torch.no_grad():
        for mode in a.modes:
            t0 = time.time()
            _, traj = po.denoise(sd, cfg, lat, mil, pe, pooled, H // 16, W // 16, N_SCHED, 30.0, max_steps=a.steps,
                                 teacher=ref if mode == "teacher_forced" else None)
            rec[mode] = [mae(traj[i][0], ref[i]) for i in range(len(traj))]
            rec[mode + "_seconds"] = round(time.time() - t0)
            print(mode, ["%.2e" % v for v in rec[mode]], flush=True)
            with open(a.out, "w") as f:
                json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
