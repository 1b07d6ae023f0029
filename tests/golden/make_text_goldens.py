#!/usr/bin/env python3
"""Golden vectors of the two text encoders, from the `transformers` installed in the build container (the reference calls
transformers==4.43.3, which is not under /root/reference; 5.15.0 is what is importable here -- version recorded in the file).

    python tests/golden/make_text_goldens.py        -> tests/golden/g10_text.safetensors

Tiny random T5 (gated-gelu, relative bias, d_kv 64) and CLIP text (quick_gelu, causal) models in fp32, fixed input ids;
stored: the state dicts, the ids and the outputs.  Data only -- no third-party source is copied."""
import os
import sys

import torch
import transformers
from safetensors.torch import save_file
from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

OUT = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(1234)
torch.set_grad_enabled(False)

T5 = dict(vocab_size=100, d_model=64, d_kv=64, d_ff=128, num_layers=2, num_heads=2, feed_forward_proj="gated-gelu",
          relative_attention_num_buckets=32, relative_attention_max_distance=128)
CLIP = dict(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
            max_position_embeddings=77, projection_dim=64, hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)

out = {"meta.transformers_version": torch.tensor([int(x) for x in transformers.__version__.split(".")[:3]])}
t5 = T5EncoderModel(T5Config(**T5)).eval()
for p in t5.parameters():           # default init leaves relative_attention_bias tiny and norms at 1: make every tensor matter
    p.add_(torch.randn_like(p) * 0.05)
ids = torch.randint(3, 100, (2, 40))
ids[1, 25:] = 0                                       # padding tokens (still attended: the pipeline passes no mask)
out["t5.ids"] = ids
out["t5.out"] = t5(ids)[0]
for k, v in t5.state_dict().items():
    out["t5.sd." + k] = v.clone()
# what the reference actually runs: from_pretrained(torch_dtype=bfloat16).  `_keep_in_fp32_modules = ["wo"]` only applies when
# the requested dtype is float16 (transformers modeling_utils: "will upcast to fp32 only if the requested dtype is fp16"; the
# pinned 4.43.3 has the same float16-only condition), so EVERY weight incl. DenseReluDense.wo is bf16 and the residual stream
# is rounded to bf16 after every sublayer.  No RNG draw here: the tensors below this line are unchanged.
import copy
t5_bf = copy.deepcopy(t5).to(torch.bfloat16)
assert t5_bf.encoder.block[0].layer[1].DenseReluDense.wo.weight.dtype == torch.bfloat16
# explicit matmul / softmax / matmul attention, the only form the pinned 4.43.3 T5 has (5.x would pick a fused SDPA kernel,
# whose bf16 rounding points are the backend's business); 5.15's eager T5 takes the softmax of the bf16 scores directly where
# 4.43.3 upcasts first -- on the CPU both round the fp32 softmax once, the same values
t5_bf.config._attn_implementation = "eager"
out["t5.out_bf16"] = t5_bf(ids)[0]
clip = CLIPTextModel(CLIPTextConfig(**CLIP)).eval()
for p in clip.parameters():
    p.add_(torch.randn_like(p) * 0.05)
cids = torch.randint(3, 119, (2, 77))
cids[:, 0] = 0
cids[0, 30] = 119                                     # the largest id marks EOS for the legacy (eos_token_id == 2) pooling
cids[0, 31:] = 1
cids[1, 76] = 119
out["clip.ids"] = cids
r = clip(cids)
out["clip.last"], out["clip.pooled"] = r.last_hidden_state, r.pooler_output
clip_bf = copy.deepcopy(clip).to(torch.bfloat16)
clip_bf.config._attn_implementation = "eager"
rb = clip_bf(cids)
out["clip.last_bf16"], out["clip.pooled_bf16"] = rb.last_hidden_state, rb.pooler_output
for k, v in clip.state_dict().items():      # on-disk key names of the pinned 4.43.3 layout (text_model.* prefix; 5.x dropped it)
    out["clip.sd." + (k if k.startswith("text_model.") else "text_model." + k)] = v.clone()
save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(OUT, "g10_text.safetensors"))
print("g10_text:", len(out), "tensors", sum(v.numel() * v.element_size() for v in out.values()) / 1e6, "MB; transformers", transformers.__version__)
