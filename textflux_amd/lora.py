"""LoRA load path: parse a diffusers/PEFT-format LoRA file and MERGE it into the fused transformer weights at load.

Reference call chain (arithmetic in third-party `peft`, absent here -- parity of that sub-path is "unpinned",
DESIGN.md): FluxLoraLoaderMixin.lora_state_dict D/loaders/lora_pipeline.py:1618-1743, load_lora_into_transformer
:1821-1861, PeftAdapterMixin.load_lora_adapter D/loaders/peft.py:111-287 (prefix strip :200-204, rank = lora_B.shape[1]
:217-220), get_peft_kwargs D/utils/peft_utils.py:150-192 (lora_alpha = alpha entry or rank).  File format
(D/loaders/lora_base.py:722-757): `transformer.<module path>.lora_A.weight [r, in]`, `.lora_B.weight [out, r]`,
optional `<module path>.alpha`.

PEFT adds B(A(x)) * (alpha / r) at every forward; this engine folds the same update into the weights once:
    W' = W + (alpha / r) * B @ A      (device GEMM through the C ABI, fp32 accumulate, one bf16 rounding of the update)
TextFlux trains alpha = r = 128 (scripts/train_lora.py:527-532), i.e. scale 1.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from . import ops

BF16 = torch.bfloat16
LORA_WEIGHT_NAME_SAFE = "pytorch_lora_weights.safetensors"


def lora_state_dict(path_or_dict, return_alphas: bool = False, weight_name: Optional[str] = None):
    if isinstance(path_or_dict, dict):
        sd = dict(path_or_dict)
    else:
        from safetensors.torch import load_file
        p = path_or_dict
        if os.path.isdir(p):
            p = os.path.join(p, weight_name or LORA_WEIGHT_NAME_SAFE)
        sd = load_file(p)
    if any("dora_scale" in k for k in sd):  # lora_pipeline.py:1705-1712: DoRA scales are dropped with a warning
        sd = {k: v for k, v in sd.items() if "dora_scale" not in k}
    alphas = {}
    for k in list(sd.keys()):
        if k.endswith(".alpha"):
            alphas[k] = sd.pop(k)
    if any(".lora_down.weight" in k or "lora_unet_" in k or ".processor." in k for k in sd):
        raise NotImplementedError("Kohya / XLabs LoRA formats are out of scope (SURVEY.md §2.2); convert to the diffusers format")
    return (sd, alphas) if return_alphas else sd


def _target_index(transformer) -> Dict[str, Tuple[str, int, int]]:
    """reference module path -> (fused tensor name, row offset, out_features)."""
    D = transformer.inner_dim
    idx = {}
    for key, name, off in transformer._fusion_map():
        rows = transformer.w[name + ".w"].shape[0]
        idx[key] = (name, off, rows)
    # out_features of a target = distance to the next offset inside the fused tensor, or the tensor's end
    by_name: Dict[str, list] = {}
    for key, (name, off, rows) in idx.items():
        by_name.setdefault(name, []).append((off, key))
    out = {}
    for name, lst in by_name.items():
        lst.sort()
        total = transformer.w[name + ".w"].shape[0]
        for i, (off, key) in enumerate(lst):
            end = lst[i + 1][0] if i + 1 < len(lst) else total
            out[key] = (name, off, end - off)
    return out


@torch.no_grad()
def merge_lora_into_transformer(state_dict: Dict[str, torch.Tensor], network_alphas: Optional[Dict[str, torch.Tensor]],
                                transformer, scale: float = 1.0) -> int:
    """Returns the number of merged target modules."""
    prefix = "transformer."
    sd = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    if not sd:
        sd = dict(state_dict)  # already stripped
    alphas = {}
    for k, v in (network_alphas or {}).items():
        k2 = k[len(prefix):] if k.startswith(prefix) else k
        alphas[k2[: -len(".alpha")]] = float(v)
    targets = _target_index(transformer)
    merged = 0
    dev = transformer.device
    modules = sorted({k[: -len(".lora_A.weight")] for k in sd if k.endswith(".lora_A.weight")})
    for mod in modules:
        if mod not in targets:
            raise KeyError(f"LoRA target {mod} is not a Linear of FluxTransformer2DModel")
        A = sd[mod + ".lora_A.weight"].to(dev, BF16)          # [r, in]
        Bm = sd[mod + ".lora_B.weight"].to(dev, BF16)         # [out, r]
        r = Bm.shape[1]
        name, off, rows = targets[mod]
        W = transformer.w[name + ".w"][off:off + rows]
        if Bm.shape[0] != rows or A.shape[1] != W.shape[1] or A.shape[0] != r:
            raise ValueError(f"LoRA shapes for {mod} do not match the target weight {tuple(W.shape)}")
        s = scale * alphas.get(mod, float(r)) / r
        gate = torch.full((W.shape[1],), s, dtype=BF16, device=dev)
        # W[out, in] += s * (B[out, r] @ A[r, in]):   C = res + gate * (a @ w^T) with a = B, w = A^T [in, r]
        ops.gemm(Bm.contiguous(), A.t().contiguous(), None, out=W, epilogue=ops.EPI_BIAS_GATE_RES, gate=gate, res=W)
        merged += 1
    transformer._session = None
    return merged
