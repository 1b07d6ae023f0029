"""The checkpoint loaders against a synthetic HF-layout directory (tests/helpers/tiny_checkpoint.py), on CPU: sharded
transformer whose weights and biases sit in different shards, fused layout == load_state_dict of the same tensors, missing
tensors detected; VAE config / key check; scheduler config."""
import json
import os

import pytest
import torch

from tests.helpers import tiny_checkpoint as tc
from textflux_amd.schedulers import FlowMatchEulerDiscreteScheduler
from textflux_amd.transformer import FluxTransformer2DModel


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("pipe"))
    sd, vsd = tc.write_pipeline_dir(root, text=False)
    return root, sd, vsd


def test_sharded_transformer_loads_like_a_flat_state_dict(ckpt):
    root, sd, _ = ckpt
    with open(os.path.join(root, "transformer", "diffusion_pytorch_model.safetensors.index.json")) as f:
        wmap = json.load(f)["weight_map"]
    split = [k[:-7] for k in wmap if k.endswith(".weight") and k[:-7] + ".bias" in wmap and wmap[k] != wmap[k[:-7] + ".bias"]]
    assert len(split) >= 50         # the layout really does separate weights from their biases
    m = FluxTransformer2DModel.from_pretrained(root, subfolder="transformer", device="cpu")
    ref = FluxTransformer2DModel.from_config(dict(num_layers=2, num_single_layers=2, num_attention_heads=2, in_channels=384,
                                                  out_channels=64, joint_attention_dim=64, pooled_projection_dim=128,
                                                  guidance_embeds=True)).load_state_dict(sd, device="cpu")
    assert m.w.keys() == ref.w.keys()
    for k in m.w:
        assert torch.equal(m.w[k], ref.w[k]), k
    # spot check of the fused layout: rows [2D, 3D) of d0.qkv_img are to_q, with its bias at the same offset
    D = m.inner_dim
    assert torch.equal(m.w["d0.qkv_img.w"][2 * D:3 * D], sd["transformer_blocks.0.attn.to_q.weight"].to(torch.bfloat16))
    assert torch.equal(m.w["d0.qkv_img.b"][2 * D:3 * D], sd["transformer_blocks.0.attn.to_q.bias"].to(torch.bfloat16))


def test_missing_tensor_is_reported(ckpt, tmp_path):
    import shutil
    from safetensors.torch import load_file, save_file
    root, _, _ = ckpt
    dst = str(tmp_path / "broken")
    shutil.copytree(os.path.join(root, "transformer"), dst)
    fn = os.path.join(dst, "diffusion_pytorch_model-00002-of-00003.safetensors")
    sh = load_file(fn)
    victim = next(k for k in sh if k.endswith(".bias"))
    del sh[victim]
    save_file(sh, fn)
    with pytest.raises(RuntimeError, match="missing 1 tensors"):
        FluxTransformer2DModel.from_pretrained(dst, device="cpu")


def test_scheduler_config_round_trip(ckpt):
    root, _, _ = ckpt
    with open(os.path.join(root, "scheduler", "scheduler_config.json")) as f:
        sch = FlowMatchEulerDiscreteScheduler.from_config(json.load(f))
    assert sch.config.use_dynamic_shifting and sch.config.max_shift == 1.15 and sch.config.shift == 3.0


def test_streamer_chunking_matches_safetensors_reader(tmp_path):
    """loader.ShardStreamer with chunk sizes from 64 B to 1 MiB (tensors cut across chunks, several tensors per chunk, dtype
    conversion on the way) against the tensors safetensors itself wrote."""
    from safetensors.torch import save_file
    from textflux_amd import loader
    g = torch.Generator().manual_seed(0)
    specs = [((3,), torch.float32), ((1000, 7), torch.bfloat16), ((5,), torch.bfloat16), ((4097, 33), torch.float32),
             ((1,), torch.float16), ((70000,), torch.bfloat16), ((16, 16), torch.float32)]
    sd = {f"t{i}": torch.randn(*shape, generator=g).to(dt) for i, (shape, dt) in enumerate(specs)}
    fn = str(tmp_path / "x.safetensors")
    save_file(sd, fn)
    for chunk in (64, 1000, 4096, 100000, 1 << 20):
        st = loader.ShardStreamer("cpu")
        st.chunk, st.bufs, st.events = chunk, [torch.empty(chunk, dtype=torch.uint8)], [None]
        out = {k: torch.empty(v.shape, dtype=torch.bfloat16) for k, v in sd.items()}
        seen = st.stream_file(fn, lambda k, shape, dt: out[k] if k != "t2" else None)      # t2 is skipped by the router
        assert set(seen) == set(sd) and st.bytes_moved > 0
        for k in sd:
            if k != "t2":
                assert torch.equal(out[k], sd[k].to(torch.bfloat16)), (chunk, k)
