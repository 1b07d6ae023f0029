"""Sweeps of the SUPERSEDED kernel variants (round 6: compiled into the bench library only -- `make -C textflux_amd/csrc bench` ->
textflux_amd/libtextflux_hip_bench.so): the 8-wave attention kernels (8 exact online maximum, 10 matrix-pipe softmax), the half-tile
pipelined kernel (20), the other bookkeeping modes of the one-wave-per-SIMD kernel (31 .. 33), the 16 x 16 x 32 kernel (40) and the
one-wave-per-SIMD GEMM (gemm_waves 4).  Not part of the driver's suite (tests/): run on a GPU box with

    TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python -m pytest tools/variant_tests -q -m gpu

(the conftest of this directory selects the bench library when TFX_LIB is unset).  The product kernels' own tests stay in tests/."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_kernels_gpu import BF, close, rnd
from tests.test_kernels_gpu import test_attention_reference_maximum_paths_vs_fp64 as _reference_maximum_paths


@pytest.fixture(scope="module")
def ops():
    from textflux_amd import _lib, ops as o
    assert _lib.LIB_PATH.endswith("_bench.so"), "these sweeps need the bench library (TFX_LIB)"
    return o


@pytest.mark.parametrize("B,M,N,K", [(1, 4096, 3072, 256), (3, 5120, 9216, 384), (2, 1000, 3136, 512), (1, 36864, 3072, 3072)])
def test_gemm_one_wave_per_simd_kernel_is_bit_identical(ops, B, M, N, K):
    """Round 4: gemm4w_kernel -- 4 waves of 128 x 128, 32-deep K sub-tiles in a 4-set LDS ring, the whole
    head of a q / k norm tile inside one wave -- accumulates every output element in the same order as the other two MFMA kernels:
    bit-identical for every epilogue, ragged edges and batch strides included.  (It measures 11-14 % slower than the ping-pong
    kernel, profiles/r04_gemm4w_ab.json, and is selectable only: tfx_set_option gemm_waves 4.)"""
    a, w = rnd((B, M, K), 71).to(BF).cuda(), rnd((N, K), 72, 0.05).to(BF).cuda()
    bias, gate, res = rnd((N,), 73).to(BF).cuda(), rnd((B, N), 74).to(BF).cuda(), rnd((B, M, N), 75).to(BF).cuda()
    cases = [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, (N // 256 - 1) * 256))),
             (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)), (ops.EPI_BIAS_RES, dict(res=res))]
    try:
        for epi, kw in cases:
            ops.set_option("gemm_waves", 8)
            ref = ops.gemm(a, w, bias, epilogue=epi, variant=3, **kw)
            ops.set_option("gemm_waves", 4)
            got = torch.full((B, M, N), 5.0, dtype=BF, device="cuda")
            ops.gemm(a, w, bias, out=got, epilogue=epi, variant=3, **kw)
            assert torch.equal(ref, got), (epi, (ref.float() - got.float()).abs().max().item())
    finally:
        ops.set_option("gemm_waves", 8)



@pytest.mark.parametrize("nw", [8, 10, 20, 30, 31, 32, 33, 40])
def test_attention_waves_variants(ops, nw):
    """The other kernels of the product library (8: exact online maximum, 10: matrix-pipe softmax with 8 waves x 32 rows, 20:
    half-tile software-pipelined) compute the same thing as the default (30: one wave per SIMD, 64 rows per wave).  The
    remaining schedules (4, 9, 12, 16) are bench-only builds."""
    B, H, N = 2, 2, 712
    q, k, v = (rnd((B, N, H * 128), s).to(BF) for s in (27, 28, 29))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * 128)
    ops.set_option("attention_waves", nw)
    try:
        got = ops.attention(q.cuda(), k.cuda(), v.cuda())
    finally:
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)
    close(got, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


@pytest.mark.parametrize("nw", [30, 31, 32, 33, 40])
@pytest.mark.parametrize("B,H,N", [(1, 1, 1), (2, 3, 8), (1, 2, 33), (1, 1, 64), (1, 1, 65), (2, 2, 96), (1, 3, 300), (2, 2, 1664), (1, 24, 520)])
def test_attention_one_wave_per_simd_kernels_shape_sweep(ops, nw, B, H, N):
    """The one-wave-per-SIMD kernels (30 / 31 / 32: 32 x 32 x 16 MFMA with the softmax bookkeeping on the matrix pipe / the row sums
    on the VALU / + the reference offset only when a row has one; 40: 16 x 16 x 32 MFMA) over the ragged / tiny-N sweep, whichever
    of them is the default."""
    q, k, v = (rnd((B, N, H * 128), s).to(BF) for s in (20, 21, 22))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * 128)
    ops.set_option("attention_waves", nw)
    try:
        got = ops.attention(q.cuda(), k.cuda(), v.cuda())
    finally:
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)
    close(got, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)



def test_attention_unaligned_output_rows_take_the_fallback_kernel(ops):
    """The default kernel stores whole rows in 16-byte pieces; an output view whose row stride is not a multiple of 8
    elements is served by the 8-wave kernel instead (same result)."""
    B, H, N = 1, 2, 300
    q, k, v = (rnd((B, N, H * 128), s).to(BF).cuda() for s in (31, 32, 33))
    a = ops.attention(q, k, v)
    buf = torch.zeros(B, N, H * 128 + 4, dtype=BF, device="cuda")
    b = ops.attention(q, k, v, out=buf[:, :, :H * 128])
    close(b, a.float().cpu().to(BF), max_rel=2e-2, mae_rel=4e-3)
    assert (buf[:, :, H * 128:] == 0).all()




@pytest.mark.parametrize("nw", [8, 10, 20, 31, 32, 33, 40])
def test_attention_reference_maximum_paths_vs_fp64_variants(ops, nw):
    """tests/test_kernels_gpu.py::test_attention_reference_maximum_paths_vs_fp64 (every path of the lazy-reference logic against an fp64
    softmax) for the kernels the product library no longer carries."""
    _reference_maximum_paths(ops, nw)


def test_lazy_reference_form_without_a_bound(ops):
    """attention_waves 34 without a score bound runs attn_w4_kernel<3> (MODE 0's stream with the lazy reference offset) here."""
    B, H, N = 2, 3, 1500
    q, k, v = (rnd((B, N, H * 128), s).to(BF) for s in (41, 42, 43))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * 128)
    try:
        ops.set_option("attention_waves", 34)
        close(ops.attention(q.cuda(), k.cuda(), v.cuda()), ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    finally:
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)


def test_whole_forward_on_the_one_wave_per_simd_gemm():
    """gemm_waves 4: every unsliced launch, the fused q / k norm + RoPE epilogue included (a head lies inside one wave instead of two),
    reproduces a 2 + 4-block full-width forward to bf16 noise (its norm sums 128 squares in another order)."""
    from oracle import flux_oracle as fo
    from oracle import pipeline_oracle as po
    from textflux_amd import ops
    from textflux_amd.transformer import FluxTransformer2DModel
    cfg = fo.FluxConfig(num_layers=2, num_single_layers=4)
    sd = fo.seeded_state_dict(cfg, 11)
    m = FluxTransformer2DModel(in_channels=384, out_channels=64, num_layers=2, num_single_layers=4, guidance_embeds=True).load_state_dict(sd, device="cuda")
    g = torch.Generator().manual_seed(3)
    S, T = 4096, 512
    kw = dict(hidden_states=torch.randn(1, S, 384, generator=g).to(BF).cuda(), encoder_hidden_states=(torch.randn(1, T, 4096, generator=g) * 0.1).to(BF).cuda(),
              pooled_projections=torch.randn(1, 768, generator=g).to(BF).cuda(), timestep=torch.tensor([0.5]).to(BF).cuda(),
              guidance=torch.tensor([30.0]).cuda(), img_ids=po.latent_image_ids(64, 64), txt_ids=torch.zeros(T, 3), return_dict=False)
    o1 = m(**kw)[0]
    ops.set_option("gemm_waves", 4)
    try:
        w4 = m(**kw)[0]
    finally:
        ops.set_option("gemm_waves", 8)
    rel4 = ((w4.float() - o1.float()).abs().mean() / o1.float().abs().mean()).item()
    assert torch.isfinite(w4.float()).all() and rel4 < 2e-2, rel4
