#!/usr/bin/env python3
"""Per-op drift budget of ONE full-depth forward + Euler step against the bf16-faithful oracle.

Setting: the g11 fixture (19 + 38 blocks, SL512 b1); teacher-forced, so every step is one forward from the ORACLE's latents.  The engine's
ops are swapped one at a time for the variant that follows the reference's op chain more literally (or merely differently), through the
library's own A/B switches -- product kernels only, nothing here is a test-only restatement:

  baseline                 the product path
  attention_waves=8        attention with the textbook exact online maximum, q NOT pre-scaled in bf16 (scale applied to the fp32 scores),
                           row sums of the fp32 weights
  attention_use_bound=0    the guarded kernel (reference maximum tracked) instead of the reference-free stream
  fuse_qk_norm_rope=0      q / k RMSNorm + RoPE as the separate pass (tfx_rmsnorm_rope) instead of the GEMM epilogue
  gemm_splitk=0            no K-sliced tiles (another fp32 summation order in the few-tile GEMMs)
  gemm_group_streams=0     text and image projections of a double block as separate launches

Compared with profiles/r05_oracle_self_noise.json: the oracle against ITSELF when only the fp32 summation order of its nn.Linear changes.

  torch_rocm_reference     NOT the engine: the oracle's plain-torch op chain (= the reference's op chain, tests/test_oracle_golden.py)
                           executed by PyTorch-ROCm on this GPU (hipBLASLt GEMMs, torch's SDPA, ATen elementwise kernels) -- what the
                           reference's OWN software stack, moved to an MI355X, gets against its CPU run.  All 30 steps.

    python tools/drift_budget.py [--steps 3]      # GPU box, ~3 min -> gpurun_out/r05_drift_budget.json
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.helpers import fulldepth as fd   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r05_drift_budget.json"))
    a = ap.parse_args()
    from safetensors.torch import load_file
    from textflux_amd import ops
    ref = load_file(fd.FIXTURE)["traj_bf16"]
    _, sd = fd.seeded_weights()
    pipe = fd.build_pipeline(sd)
    del sd
    tr = pipe.transformer
    mae = lambda x, y: (x.float().cpu() - y.float().cpu()).abs().mean().item()

    def run():
        got = []

        def cb(p, i, t, k):
            got.append(k["latents"][0].float().cpu())
            if len(got) == a.steps:
                p._interrupt = True
            return {"latents": ref[i][None].cuda()}
        pipe(callback_on_step_end=cb, **fd.call_kwargs())
        return [mae(got[i], ref[i]) for i in range(a.steps)]

    def opt(name, val, restore):
        def f(on):
            ops.set_option(name, val if on else restore)
        return f

    def attr(name, val):
        def f(on):
            setattr(tr, name, val if on else not val)
            tr._session = None
        return f

    variants = [("baseline", None), ("attention_waves=8", opt("attention_waves", 8, ops.DEFAULT_ATTENTION)),
                ("attention_use_bound=0", opt("attention_use_bound", 0, 1)), ("fuse_qk_norm_rope=0", attr("fuse_qk_norm_rope", False)),
                ("gemm_splitk=0", opt("gemm_splitk", 0, 1)), ("gemm_group_streams=0", opt("gemm_group_streams", 0, 1))]
    rec = {"what": "teacher-forced latent MAE of the first steps, full 19+38 model SL512 b1 (g11 fixture), one engine op swapped at a time",
           "steps": a.steps, "rows": {}}
    for name, sw in variants:
        if sw:
            sw(True)
        try:
            rec["rows"][name] = run()
        finally:
            if sw:
                sw(False)
        print(name, ["%.3e" % e for e in rec["rows"][name]], flush=True)
    # engine, all 30 steps (what tests/test_fulldepth_trajectory_gpu.py asserts)
    a.steps = fd.N_SCHED
    rec["engine_all_steps"] = run()
    print("engine_all_steps", ["%.3e" % e for e in rec["engine_all_steps"]], flush=True)
    # ---- the reference's op chain on PyTorch-ROCm (checker code on the GPU; nothing of this is product path)
    del pipe, tr
    torch.cuda.empty_cache()
    from oracle import pipeline_oracle as po
    cfg, sd = fd.seeded_weights()
    sd = {k: v.cuda() for k, v in sd.items()}
    lat, mil, pe, pooled = (t.cuda() for t in fd.inputs())
    from oracle import flux_oracle as fo
    def on_gpu(**kw):
        with torch.device("cuda"):                   # factory calls inside the forward (arange, zeros) land on the GPU
            return fo.transformer_forward(sd, cfg, **{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()})

    with torch.no_grad():                            # scheduler tables stay host tensors, as in the reference's pipeline
        _, traj = po.denoise(sd, cfg, lat, mil, pe, pooled, fd.H // 16, fd.W // 16, fd.N_SCHED, 30.0, teacher=[r.cuda() for r in ref],
                             model_fn=on_gpu)
    rec["torch_rocm_reference_all_steps"] = [mae(traj[i][0], ref[i]) for i in range(fd.N_SCHED)]
    print("torch_rocm_reference_all_steps", ["%.3e" % e for e in rec["torch_rocm_reference_all_steps"]], flush=True)
    rec["torch"] = torch.__version__
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
