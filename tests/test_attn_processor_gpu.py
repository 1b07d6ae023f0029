"""Plugin point 3: textflux_amd.attn_processor.TfxFluxAttnProcessor called the way the reference's
Attention module calls its processor, with a stand-in `attn` module (torch.nn Linears + RMSNorm weights = what
diffusers' Attention holds for FLUX), against the oracle's FluxAttnProcessor2_0 restatement -- double-stream (text + image)
and single-stream (pre_only) forms."""
from types import SimpleNamespace

import pytest
import torch

from oracle import flux_oracle as fo
from oracle import pipeline_oracle as po

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class StandInAttention(torch.nn.Module):
    def __init__(self, sd, prefix, heads, context):
        super().__init__()
        self.heads = heads

        def lin(name):
            w, b = sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"]
            m = torch.nn.Linear(w.shape[1], w.shape[0])
            m.weight.data, m.bias.data = w.clone(), b.clone()
            return m

        self.to_q, self.to_k, self.to_v = lin("to_q"), lin("to_k"), lin("to_v")
        self.norm_q = SimpleNamespace(weight=sd[f"{prefix}.norm_q.weight"].to(BF).cuda())
        self.norm_k = SimpleNamespace(weight=sd[f"{prefix}.norm_k.weight"].to(BF).cuda())
        if context:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = lin("add_q_proj"), lin("add_k_proj"), lin("add_v_proj")
            self.to_out = torch.nn.ModuleList([lin("to_out.0")])
            self.to_add_out = lin("to_add_out")
            self.norm_added_q = SimpleNamespace(weight=sd[f"{prefix}.norm_added_q.weight"].to(BF).cuda())
            self.norm_added_k = SimpleNamespace(weight=sd[f"{prefix}.norm_added_k.weight"].to(BF).cuda())


def rel_mae(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().mean() / b.abs().mean()).item()


@torch.no_grad()
def test_processor_behind_a_reference_style_attention_module():
    from textflux_amd.attn_processor import TfxFluxAttnProcessor
    cfg = fo.FluxConfig(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    sd = fo.seeded_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(0)
    B, S, T, D = 2, 96, 40, 256
    hidden, enc = torch.randn(B, S, D, generator=g), torch.randn(B, T, D, generator=g)
    cos, sin = fo.flux_pos_embed(torch.cat([torch.zeros(T, 3), po.latent_image_ids(8, 12)]))
    proc = TfxFluxAttnProcessor()
    # double-stream block's attention: returns (image stream, text stream), both through their output projections
    attn = StandInAttention(sd, "transformer_blocks.0.attn", 2, context=True).to("cuda", BF)
    o_img, o_txt = proc(attn, hidden.to(BF).cuda(), encoder_hidden_states=enc.to(BF).cuda(), image_rotary_emb=(cos, sin))
    r_img, r_txt = fo.flux_attention(sd, "transformer_blocks.0.attn", 2, hidden, enc, cos, sin)
    assert o_img.shape == r_img.shape and o_txt.shape == r_txt.shape
    assert rel_mae(o_img, r_img) < 1.5e-2 and rel_mae(o_txt, r_txt) < 1.5e-2
    # single-stream block's attention (pre_only: no output projection, joint sequence in)
    attn1 = StandInAttention(sd, "single_transformer_blocks.0.attn", 2, context=False).to("cuda", BF)
    joint = torch.cat([enc, hidden], 1)
    o1 = proc(attn1, joint.to(BF).cuda(), image_rotary_emb=(cos, sin))
    r1 = fo.flux_attention(sd, "single_transformer_blocks.0.attn", 2, joint, None, cos, sin)
    assert o1.shape == r1.shape and rel_mae(o1, r1) < 1.5e-2
    with pytest.raises(NotImplementedError):
        proc(attn1, joint.to(BF).cuda(), attention_mask=torch.ones(1), image_rotary_emb=(cos, sin))
