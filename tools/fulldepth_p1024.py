#!/usr/bin/env python3
"""Full depth (19 + 38 blocks) at the HEADLINE geometry (P1024: 1024 x 1024, S = 4096, N = 4608; what bench.py is quoted on), one forward +
Euler step at three points of the 30-step schedule (first, middle, last), each from seeded latents -- any latents are a valid input to a
teacher-forced step, so no 30-step CPU trajectory is needed: the bf16-faithful oracle (bit-exact restatement of the reference's bf16 run)
and, for the floor, the same oracle with the fp32 summation order of every nn.Linear permuted (tools/oracle_self_noise.py).

    python tools/fulldepth_p1024.py          # CPU, build container, ~10 min on 8 cores -> tests/golden/g14_fulldepth_p1024_oracle.safetensors

Consumed by tests/test_fulldepth_trajectory_gpu.py::test_headline_geometry_full_depth_steps.  Weights: tests/helpers/fulldepth.seeded_weights
(the g11 model).  Reference: D/pipelines/flux/pipeline_flux_fill.py:2053-2112, D/models/transformers/transformer_flux.py:1028-1212.
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.helpers import fulldepth as fd   # noqa: E402


def main():
    from safetensors.torch import save_file
    from oracle import flux_oracle as fo
    from oracle import pipeline_oracle as po
    torch.set_num_threads(int(os.environ.get("THREADS", os.cpu_count() or 1)))
    cfg, sd = fd.seeded_weights()
    out = {}
    mae = lambda x, y: (x.float() - y.float()).abs().mean().item()
    with torch.no_grad():
        for tag in ("oracle", "permuted"):
            if tag == "permuted":
                perms = {}

                def perm(k):
                    if k not in perms:
                        perms[k] = torch.randperm(k, generator=torch.Generator().manual_seed(k))
                    return perms[k]
                for name in list(sd):
                    if name.endswith(".weight") and sd[name].dim() == 2:
                        sd[name] = sd[name][:, perm(sd[name].shape[1])].contiguous()
                plain = fo.linear
                fo.linear = lambda x, sd_, name: plain(x[..., perm(x.shape[-1])], sd_, name)
            for k in fd.P1024_STEPS:
                lat, mil, pe, pooled = fd.p1024_inputs(k)
                t0 = time.time()
                # teacher-forced step k: denoise() walks the schedule from 0, `teacher` supplies the latents step k starts from
                teacher = [lat] * fd.N_SCHED
                fwd_count = {"n": 0}
                real = fo.transformer_forward

                def only_step_k(**kw):
                    i = fwd_count["n"]
                    fwd_count["n"] += 1
                    if i != k:          # the other steps' forwards are not needed: their result is overwritten by the teacher
                        return torch.zeros(1, fd.P1024_S, 64, dtype=lat.dtype)
                    return real(sd, cfg, **kw)
                _, traj = po.denoise(sd, cfg, lat, mil, pe, pooled, 64, 64, fd.N_SCHED, 30.0, teacher=teacher, model_fn=only_step_k,
                                     max_steps=k + 1)
                out[f"{tag}.step{k}"] = traj[k][0].to(torch.bfloat16).contiguous()
                print(tag, k, f"{time.time() - t0:.0f} s", flush=True)
    floor = {str(k): mae(out[f"permuted.step{k}"], out[f"oracle.step{k}"]) for k in fd.P1024_STEPS}
    print("self-noise floor", floor)
    keep = {k: v for k, v in out.items() if k.startswith("oracle.")}
    save_file(keep, fd.P1024_FIXTURE, metadata={"floor": json.dumps(floor), "tool": "tools/fulldepth_p1024.py"})


if __name__ == "__main__":
    main()
