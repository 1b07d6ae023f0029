"""Writes a tiny FLUX.1-Fill pipeline directory in the HF layout the reference's loaders read:

    <root>/model_index.json
    <root>/transformer/{config.json, diffusion_pytorch_model-0000k-of-00003.safetensors, ...index.json}   (sharded)
    <root>/vae/{config.json, diffusion_pytorch_model.safetensors}
    <root>/scheduler/scheduler_config.json
    <root>/text_encoder/ (CLIPTextModel)   <root>/tokenizer/   (byte-level BPE without merges)
    <root>/text_encoder_2/ (T5EncoderModel) <root>/tokenizer_2/ (unigram over printable ASCII)
    <lora>/pytorch_lora_weights.safetensors                                                       (a18 key format)

All weights are the seeded synthetic ones of the oracle (no hub, no real checkpoints offline).  Shard k of the transformer
holds every key whose position in state-dict order is k mod 3, so practically every Linear has its weight and its bias in
DIFFERENT shards (the loader must not assume they travel together)."""
import json
import os

import torch
from safetensors.torch import save_file

from oracle import flux_oracle as fo
from oracle import vae_oracle as vo

TR_CFG = fo.FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=128)
VAE_KW = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=16, norm_num_groups=16)
SCHED = dict(_class_name="FlowMatchEulerDiscreteScheduler", num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True,
             base_shift=0.5, max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096)
LORA_TARGETS = ["transformer_blocks.0.attn.to_q", "transformer_blocks.1.attn.add_k_proj", "transformer_blocks.0.ff.net.0.proj",
                "transformer_blocks.1.ff_context.net.2", "single_transformer_blocks.0.attn.to_v",
                "single_transformer_blocks.1.attn.to_k"]


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def write_tokenizers(root):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors
    from transformers import CLIPTokenizer, PreTrainedTokenizerFast
    chars = list(_bytes_to_unicode().values())
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    d = os.path.join(root, "tokenizer")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"), model_max_length=77).save_pretrained(d)
    pieces = ([("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0), ("▁", -2.0)] + [(chr(c), -3.0) for c in range(33, 127)] +
              [("▁" + chr(c), -2.5) for c in range(97, 123)])
    tk = Tokenizer(models.Unigram(pieces, unk_id=2))
    tk.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁")
    tk.decoder = decoders.Metaspace(replacement="▁")
    tk.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    d2 = os.path.join(root, "tokenizer_2")
    os.makedirs(d2, exist_ok=True)
    PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", eos_token="</s>", unk_token="<unk>",
                            model_max_length=512).save_pretrained(d2)
    return len(vocab), len(pieces)


def write_text_encoders(root, clip_vocab, t5_vocab, seed=0):
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    torch.manual_seed(seed)
    # heads of dim 64 in both (the real CLIP-L / T5-XXL geometry the HIP attention kernel is written for); CLIP pools at the
    # largest token id (eos_token_id = 2: the legacy convention of FLUX's text_encoder config)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=clip_vocab, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                        num_attention_heads=2, max_position_embeddings=77, projection_dim=64,
                                        hidden_act="quick_gelu", bos_token_id=clip_vocab - 2, eos_token_id=2,
                                        pad_token_id=clip_vocab - 1)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=t5_vocab, d_model=64, d_kv=64, d_ff=128, num_layers=2, num_heads=2,
                                 feed_forward_proj="gated-gelu", relative_attention_num_buckets=32,
                                 relative_attention_max_distance=128)).eval()
    for m in (clip, t5):
        for prm in m.parameters():
            prm.data.add_(torch.randn_like(prm) * 0.05)
    clip.save_pretrained(os.path.join(root, "text_encoder"), safe_serialization=True)
    t5.save_pretrained(os.path.join(root, "text_encoder_2"), safe_serialization=True)
    return clip, t5


def write_pipeline_dir(root, seed_tr=7, seed_vae=900, text=True):
    """Returns (transformer state dict, vae state dict) as written (fp32 masters; files hold bf16)."""
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "model_index.json"), "w") as f:
        json.dump({"_class_name": "FluxFillPipeline", "_diffusers_version": "0.32.0.dev0",
                   "scheduler": ["diffusers", "FlowMatchEulerDiscreteScheduler"], "transformer": ["diffusers", "FluxTransformer2DModel"],
                   "vae": ["diffusers", "AutoencoderKL"], "text_encoder": ["transformers", "CLIPTextModel"],
                   "text_encoder_2": ["transformers", "T5EncoderModel"], "tokenizer": ["transformers", "CLIPTokenizer"],
                   "tokenizer_2": ["transformers", "T5TokenizerFast"]}, f)
    # ---- transformer, sharded so that weights and biases are split up
    sd = fo.seeded_state_dict(TR_CFG, seed_tr)
    td = os.path.join(root, "transformer")
    os.makedirs(td, exist_ok=True)
    c = TR_CFG
    with open(os.path.join(td, "config.json"), "w") as f:
        json.dump(dict(_class_name="FluxTransformer2DModel", patch_size=1, in_channels=c.in_channels, out_channels=c.out_channels,
                       num_layers=c.num_layers, num_single_layers=c.num_single_layers, attention_head_dim=128,
                       num_attention_heads=c.num_attention_heads, joint_attention_dim=c.joint_attention_dim,
                       pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=True, axes_dims_rope=[16, 56, 56]), f)
    shards, wmap = [dict(), dict(), dict()], {}
    for i, (k, v) in enumerate(sd.items()):
        fn = f"diffusion_pytorch_model-{i % 3 + 1:05d}-of-00003.safetensors"
        shards[i % 3][k] = v.to(torch.bfloat16).contiguous()
        wmap[k] = fn
    for j, sh in enumerate(shards):
        save_file(sh, os.path.join(td, f"diffusion_pytorch_model-{j + 1:05d}-of-00003.safetensors"))
    with open(os.path.join(td, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": sum(v.numel() * 2 for v in sd.values())}, "weight_map": wmap}, f)
    # ---- VAE
    vcfg = vo.VaeConfig(**VAE_KW)
    vsd = vo.seeded_state_dict(vcfg, seed_vae)
    vd = os.path.join(root, "vae")
    os.makedirs(vd, exist_ok=True)
    with open(os.path.join(vd, "config.json"), "w") as f:
        json.dump(dict(_class_name="AutoencoderKL", in_channels=3, out_channels=3, block_out_channels=list(VAE_KW["block_out_channels"]),
                       layers_per_block=1, latent_channels=16, norm_num_groups=16, scaling_factor=0.3611, shift_factor=0.1159,
                       use_quant_conv=False, use_post_quant_conv=False), f)
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in vsd.items()}, os.path.join(vd, "diffusion_pytorch_model.safetensors"))
    os.makedirs(os.path.join(root, "scheduler"), exist_ok=True)
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(SCHED, f)
    if text:
        cv, tv = write_tokenizers(root)
        write_text_encoders(root, cv, tv)
    return sd, vsd


def write_lora(lora_dir, sd, rank=8, seed=3):
    """LoRA file in the reference's format (D/loaders/lora_base.py:722-757); returns the merged fp32 state dict."""
    os.makedirs(lora_dir, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    lora, merged = {}, dict(sd)
    BF = torch.bfloat16
    for i, t in enumerate(LORA_TARGETS):
        out_f, in_f = sd[t + ".weight"].shape
        A = (torch.randn(rank, in_f, generator=g) * 0.2).to(BF)
        Bm = (torch.randn(out_f, rank, generator=g) * 0.2).to(BF)
        alpha = float(rank) if i % 2 == 0 else 4.0
        lora[f"transformer.{t}.lora_A.weight"], lora[f"transformer.{t}.lora_B.weight"] = A, Bm
        if i % 2:
            lora[f"transformer.{t}.alpha"] = torch.tensor(alpha)
        merged[t + ".weight"] = sd[t + ".weight"].to(BF).float() + ((alpha / rank) * (Bm.float() @ A.float()).to(BF).float()).to(BF).float()
    save_file(lora, os.path.join(lora_dir, "pytorch_lora_weights.safetensors"))
    return merged
