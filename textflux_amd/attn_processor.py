"""Plugin point 3 of the reference: an attention-processor object for the reference's OWN
`diffusers.models.attention_processor.Attention` module -- `model.set_attn_processor(TfxFluxAttnProcessor())`
(D/models/transformers/transformer_flux.py:950-982).  Same call contract as FluxAttnProcessor2_0
(D/models/attention_processor.py:1979-2060): keeps the module's projections, runs the per-head RMSNorm + RoPE and the joint
attention through the C ABI (tfx_rmsnorm_rope, tfx_joint_attention).  This is the narrowest way to put the HIP kernels
behind an unmodified reference model; the engine's own path (tfx_dit_forward) fuses far more.  Executed by
tests/test_attn_processor_gpu.py with a stand-in `attn` module."""
from __future__ import annotations

import torch

from . import ops


class TfxFluxAttnProcessor:
    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                 attention_mask=None, image_rotary_emb=None):
        if attention_mask is not None:
            raise NotImplementedError("FLUX attention is unmasked (the reference never passes a mask on this path)")
        B, S, D = hidden_states.shape
        H = attn.heads
        if D // H != 128:
            raise ValueError("the gfx950 attention kernel is specialised for head_dim 128 (FLUX.1)")
        # fused buffer [k | v | q], text rows first -- the layout tfx_rmsnorm_rope / tfx_joint_attention expect
        img = torch.cat([attn.to_k(hidden_states), attn.to_v(hidden_states), attn.to_q(hidden_states)], -1)
        if encoder_hidden_states is not None:
            e = encoder_hidden_states
            txt = torch.cat([attn.add_k_proj(e), attn.add_v_proj(e), attn.add_q_proj(e)], -1)
            y, T = torch.cat([txt, img], 1).contiguous(), e.shape[1]
            wq_t, wk_t = attn.norm_added_q.weight, attn.norm_added_k.weight
        else:
            y, T = img.contiguous(), 0
            wq_t, wk_t = attn.norm_q.weight, attn.norm_k.weight
        cos, sin = image_rotary_emb
        dev = y.device
        ops.rmsnorm_rope_(y, 2 * D, 0, H, T, attn.norm_q.weight, attn.norm_k.weight, wq_t, wk_t,
                          cos.to(dev, torch.float32).contiguous(), sin.to(dev, torch.float32).contiguous(), eps=1e-6)
        out = ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D])
        if encoder_hidden_states is None:
            return out
        return attn.to_out[0](out[:, T:]), attn.to_add_out(out[:, :T])
