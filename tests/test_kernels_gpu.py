"""GPU parity tests of the individual HIP kernels (through the C ABI) against the CPU oracle / golden vectors.

Tolerances: bf16 has 8 bits of mantissa, so one rounding is a relative error of 2^-9 = 0.2 %.  Kernels that
reproduce the reference's rounding chain are compared bit-exactly or to 1 bf16 ulp; GEMM / attention outputs are
compared with the fp32 oracle evaluated on the same bf16 inputs: |err| <= 1e-2 * max|ref| (max-norm) and a mean
absolute error <= 2e-3 * mean|ref|.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_oracle as fo
from oracle import pipeline_oracle as po
from oracle import sched_oracle as so

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from textflux_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, max_rel=1e-2, mae_rel=2e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert err.max().item() <= max_rel * ref.abs().max().item() + 1e-6, (err.max().item(), ref.abs().max().item())
    assert err.mean().item() <= mae_rel * ref.abs().mean().item() + 1e-7, (err.mean().item(), ref.abs().mean().item())


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,batch", [(256, 256, 64, 1), (300, 264, 128, 2), (37, 72, 192, 3), (1024, 768, 3072, 1),
                                         (520, 3072, 384, 2), (8, 3072, 256, 1), (4, 64, 64, 1)])
@pytest.mark.parametrize("variant", [0, 1, 2])   # 0 generic FMA, 1 MFMA (persistent kernel when K % 128 == 0), 2 one-tile MFMA
def test_gemm_bias(ops, M, N, K, batch, variant):
    a, w, b = rnd((batch, M, K), 1).to(BF), rnd((N, K), 2, 0.05).to(BF), rnd((N,), 3).to(BF)
    ref = a.float() @ w.float().T + b.float()
    got = ops.gemm(a.cuda(), w.cuda(), b.cuda(), variant=variant)
    close(got, ref.to(BF))


def test_gemm_generic_odd_k(ops):
    a, w, b = rnd((2, 19, 32), 1).to(BF), rnd((24, 32), 2).to(BF), rnd((24,), 3).to(BF)
    close(ops.gemm(a.cuda(), w.cuda(), b.cuda()), (a.float() @ w.float().T + b.float()).to(BF))


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_gemm_gelu_split_and_strided(ops, variant):
    """The single-block fused projection: plain bias below column 256, tanh-GELU from column 256 on; A and C are
    column slices of wider buffers (lda/ldc > row length)."""
    B, M, K, N = 2, 130, 128, 512
    abuf = rnd((B, M, K + 64), 4).to(BF).cuda()
    a = abuf[:, :, 64:]
    w, b = rnd((N, K), 5, 0.1).to(BF), rnd((N,), 6).to(BF)
    cbuf = torch.zeros(B, M, N + 128, dtype=BF, device="cuda")
    out = cbuf[:, :, 128:]
    ops.gemm(a, w.cuda(), b.cuda(), out=out, epilogue=ops.EPI_BIAS_GELU, gelu_from_col=256, variant=variant)
    lin = a.float().cpu() @ w.float().T + b.float()
    ref = torch.cat([lin[..., :256], torch.nn.functional.gelu(lin[..., 256:], approximate="tanh")], -1)
    close(out, ref.to(BF))
    assert cbuf[:, :, :128].abs().max().item() == 0  # nothing written outside the slice


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_gemm_gate_residual_inplace(ops, variant):
    B, M, K, N = 2, 200, 256, 256
    a, w, b = rnd((B, M, K), 7).to(BF), rnd((N, K), 8, 0.05).to(BF), rnd((N,), 9).to(BF)
    gate, res = rnd((B, N), 10).to(BF), rnd((B, M, N), 11).to(BF)
    lin = (a.float() @ w.float().T + b.float()).to(BF)
    ref = res.float() + (gate.float()[:, None] * lin.float()).to(BF).float()
    h = res.clone().cuda()
    ops.gemm(a.cuda(), w.cuda(), b.cuda(), out=h, epilogue=ops.EPI_BIAS_GATE_RES, gate=gate.cuda(), res=h, variant=variant)
    close(h, ref.to(BF))


def test_gemm_fast_matches_generic_on_device(ops):
    """Asymmetric data, long K: the MFMA kernel against the fp32-FMA kernel on the same device inputs."""
    a = (rnd((3, 777, 3072), 12) + 0.3).to(BF).cuda()
    w = (rnd((1032, 3072), 13, 0.03) - 0.01).to(BF).cuda()
    b = rnd((1032,), 14).to(BF).cuda()
    close(ops.gemm(a, w, b, variant=1), ops.gemm(a, w, b, variant=0), max_rel=5e-3, mae_rel=1e-3)


@pytest.mark.parametrize("B,M,N,K", [(1, 36864, 3072, 256), (3, 5120, 9216, 384), (2, 5248, 3072, 384), (2, 1000, 3136, 128),
                                     (1, 512, 3072, 3072), (1, 8, 64, 128), (3, 4736, 3328, 256)])
def test_gemm_persistent_bit_identical_to_one_tile_kernel(ops, B, M, N, K):
    """The persistent kernel (variant 3; many tiles per block, next-tile prefetch, chunked epilogue, both request
    placements) accumulates in the same order as the one-tile kernel (variant 2): every epilogue must agree bit for bit,
    including ragged M / N edges, batch strides and grids smaller and larger than the CU count."""
    a, w = rnd((B, M, K), 21).to(BF).cuda(), rnd((N, K), 22, 0.05).to(BF).cuda()
    bias, gate, res = rnd((N,), 23).to(BF).cuda(), rnd((B, N), 24).to(BF).cuda(), rnd((B, M, N), 25).to(BF).cuda()
    cases = [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, (N // 256 - 1) * 256))),
             (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)), (ops.EPI_BIAS_RES, dict(res=res))]
    try:
        for epi, kw in cases:
            one = torch.full((B, M, N), 7.0, dtype=BF, device="cuda")
            ops.gemm(a, w, bias, out=one, epilogue=epi, variant=2, **kw)
            for place in (1, 2):
                ops.set_option("gemm_place", place)
                per = torch.full((B, M, N), 9.0, dtype=BF, device="cuda")
                ops.gemm(a, w, bias, out=per, epilogue=epi, variant=3, **kw)
                assert torch.equal(one, per), (epi, place, (one.float() - per.float()).abs().max().item())
    finally:
        ops.set_option("gemm_place", 2)


@pytest.mark.parametrize("B,M,N,K", [(1, 1664, 3072, 3072), (1, 1152, 3072, 12288), (2, 300, 3136, 4096), (1, 512, 9216, 3072)])
def test_gemm_split_k_matches_unsplit(ops, B, M, N, K):
    """Few-tile GEMMs (fewer 256x256 tiles than CUs) with a workspace take the split-K path: fp32 partials per K slice,
    summed in slice order by the second pass with the same epilogues.  Against the unsplit persistent kernel the only
    difference is the fp32 summation order: bf16 results agree except for isolated one-ulp flips; without a workspace,
    or with gemm_splitk = 0, the auto path is bit-identical to the unsplit kernel."""
    a, w = rnd((B, M, K), 51).to(BF).cuda(), rnd((N, K), 52, 0.03).to(BF).cuda()
    bias, gate, res = rnd((N,), 53).to(BF).cuda(), rnd((B, N), 54).to(BF).cuda(), rnd((B, M, N), 55).to(BF).cuda()
    ws = torch.empty(4 * B * M * N, dtype=torch.float32, device="cuda")
    cases = [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, (N // 256 - 2) * 256))),
             (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)), (ops.EPI_BIAS_RES, dict(res=res))]
    for epi, kw in cases:
        unsplit = ops.gemm(a, w, bias, epilogue=epi, variant=3, **kw)
        split = ops.gemm(a, w, bias, epilogue=epi, variant=1, workspace=ws, **kw)
        d = (unsplit.float() - split.float()).abs()
        assert torch.isfinite(split).all()
        assert d.max().item() <= 2 ** -7 * unsplit.float().abs().max().item()          # at most one bf16 ulp of the largest value
        assert (d > 0).float().mean().item() < 2e-2 and d.mean().item() < 1e-5 * max(1.0, unsplit.float().abs().mean().item())
        assert torch.equal(ops.gemm(a, w, bias, epilogue=epi, variant=1, **kw), unsplit)            # no workspace: no split
    ops.set_option("gemm_splitk", 0)
    try:
        assert torch.equal(ops.gemm(a, w, bias, variant=1, workspace=ws), ops.gemm(a, w, bias, variant=3))
    finally:
        ops.set_option("gemm_splitk", 2)


@pytest.mark.parametrize("B,M,N,K", [(1, 4096, 9216, 3072), (2, 2048, 9216, 3072), (1, 1664, 21504, 3072), (1, 1280, 15360, 3072)])
def test_gemm_tail_split_matches_unsplit(ops, B, M, N, K):
    """Round 4: a SMALL GEMM (fewer than 8 rounds of the chip) whose last round of 256 x 256 tiles is partly filled has the tiles of
    that round K-sliced (the last r tile positions of EVERY batch sample, so that identical samples keep identical bits; fp32
    partials, tail_reduce_kernel) -- 576 tiles = 2 rounds + 64 tiles in 4 slices, 588 = 2 rounds + 76 in 3 (ragged M), 300 = 1
    round + 44 in 4.  Same contract as the whole-GEMM split: against the unsplit persistent kernel only the fp32 summation order
    of the sliced tiles differs; `gemm_splitk` 1 (round 3's behaviour) and 0 leave these shapes unsplit, bit for bit."""
    a, w = rnd((B, M, K), 61).to(BF).cuda(), rnd((N, K), 62, 0.03).to(BF).cuda()
    bias, gate, res = rnd((N,), 63).to(BF).cuda(), rnd((B, N), 64).to(BF).cuda(), rnd((B, M, N), 65).to(BF).cuda()
    if B > 1:
        a[1], res[1], gate[1] = a[0], res[0], gate[0]
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")        # the engine's scratch size: 256 units of 256 KiB
    cases = [(ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=max(0, (N // 256 - 2) * 256))),
             (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)), (ops.EPI_BIAS_RES, dict(res=res))]
    for epi, kw in cases:
        unsplit = ops.gemm(a, w, bias, epilogue=epi, variant=3, **kw)
        split = ops.gemm(a, w, bias, epilogue=epi, variant=1, workspace=ws, **kw)
        d = (unsplit.float() - split.float()).abs()
        assert torch.isfinite(split).all()
        assert d.max().item() <= 2 ** -7 * unsplit.float().abs().max().item()
        frac = (d > 0).float().mean().item()
        assert 0 < frac < 2e-2, (epi, frac)            # something WAS sliced (isolated one-ulp flips), and only a sliver
        if B > 1:
            assert torch.equal(split[0], split[1])     # identical samples: identical bits (same tile positions sliced in both)
        for lvl in (1, 0):
            ops.set_option("gemm_splitk", lvl)
            try:
                assert torch.equal(ops.gemm(a, w, bias, epilogue=epi, variant=1, workspace=ws, **kw), unsplit)
            finally:
                ops.set_option("gemm_splitk", 2)


@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_qkn_entry_point_matches_gemm_plus_separate_norm_rope_pass(ops, gelu):
    """tfx_gemm_bf16_qkn (round 6, ABI 7): the fused [k | v | q (| mlp)] projection -- per-head RMSNorm + RoPE of the q / k column ranges in
    the GEMM's epilogue, the kernel tfx_dit_forward launches -- against tfx_gemm_bf16 followed by tfx_rmsnorm_rope on the same operands:
    v (and the GELU'd mlp) columns bit-identical, q / k columns within one bf16 step of their rotation pair on all but a sliver of elements
    (same rounding points, another fp32 summation order of the 128 squares); position offset pos0; shapes it cannot take are refused."""
    D, M, K, pos0 = 3072, 2304 + 40, 512, 7                  # 10 row tiles (the last ragged) x 36 / 48 column tiles >= 256 CUs
    N = 4 * D if gelu else 3 * D
    a, w, b = rnd((M, K), 81).to(BF).cuda(), rnd((N, K), 82, 0.05).to(BF).cuda(), rnd((N,), 83).to(BF).cuda()
    wq, wk = (1 + 0.1 * rnd((128,), 84)).to(BF).cuda(), (1 + 0.1 * rnd((128,), 85)).to(BF).cuda()
    ang = rnd((M + pos0, 64), 86) * 3.0
    cs = torch.stack([torch.cos(ang), torch.sin(ang)], -1).contiguous().cuda()                       # [rows, 64, 2] fp32
    cos = torch.cos(ang).repeat_interleave(2, 1).contiguous().cuda()[pos0:]
    sin = torch.sin(ang).repeat_interleave(2, 1).contiguous().cuda()[pos0:]
    kw = dict(epilogue=ops.EPI_BIAS_GELU, gelu_from_col=3 * D) if gelu else dict(epilogue=ops.EPI_BIAS)
    sep = ops.gemm(a, w, b, **kw)
    ops.rmsnorm_rope_(sep.view(1, M, N), 2 * D, 0, 24, 0, wq, wk, wq, wk, cos, sin)
    fused = ops.gemm_qkn(a, w, b, wq, wk, cs, (2 * D, 3 * D), (0, D), pos0=pos0, **kw)
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(fused[:, D:2 * D], sep[:, D:2 * D]) and torch.equal(fused[:, 3 * D:], sep[:, 3 * D:])
    for lo in (0, 2 * D):
        x, y = sep[:, lo:lo + D].float(), fused[:, lo:lo + D].float()
        diff = (x - y).abs()
        pair = (x.view(M, -1, 2) ** 2).sum(-1).sqrt().repeat_interleave(2, dim=-1)
        assert (diff <= 2 ** -6 * pair + 1e-6).all(), (lo, (diff / (2 ** -6 * pair + 1e-6)).max().item())
        assert (diff > 0).float().mean().item() < 2e-2
    assert torch.equal(ops.gemm_qkn(a, w, b, wq, wk, cs, (2 * D, 3 * D), (0, D), pos0=pos0, **kw), fused)     # deterministic
    with pytest.raises(RuntimeError, match="not eligible|gemm_qkn"):
        ops.gemm_qkn(a, w, b, wq, wk, cs, (2 * D + 128, 3 * D), (0, D), pos0=pos0, **kw)              # not whole head pairs


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_gemm_per_batch_weights_equal_one_launch_per_sample(ops, variant):
    """tfx_gemm_args.w_bstride (round 6): W [B, N, K], a different weight matrix per batch sample -- what the VAE mid-block attention's q k^T
    and P v products are -- in ONE launch, bit-identical to a launch per sample (same tiles, same kernel), in every kernel form (0 generic,
    1 auto incl. K-sliced units, 2 one-tile, 3 persistent) and for the fp32-output entry point."""
    B, M, N, K = 3, 520, 392, 256
    a, w, bias = rnd((B, M, K), 91).to(BF).cuda(), rnd((B, N, K), 92, 0.05).to(BF).cuda(), rnd((N,), 93).to(BF).cuda()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda") if variant == 1 else None
    got = ops.gemm(a, w, bias, variant=variant, workspace=ws)
    for b in range(B):
        assert torch.equal(got[b], ops.gemm(a[b], w[b], bias, variant=variant, workspace=ws)), b
    close(got, (a.float() @ w.float().transpose(1, 2) + bias.float()).to(BF))
    f = ops.gemm_f32(a, w)
    for b in range(B):
        assert torch.equal(f[b], ops.gemm_f32(a[b], w[b])), b
    wv = torch.zeros(B, N, K + 64, dtype=BF, device="cuda")[:, :, :K]          # strided views: row pitch and batch pitch of their own
    wv.copy_(w)
    assert torch.equal(ops.gemm(a, wv, bias, variant=variant, workspace=ws), got)


def test_mfma_peak_probe_reports_a_plausible_matrix_pipe_rate(ops):
    """tfx_mfma_peak_probe (round 6): the product library's MFMA-only kernel, the live denominator of bench.py's roofline.frac_of_capped.
    Its FLOP count is 64 (e4m3: 32) MFMAs per wave and K-tile x 8 waves x CUs; the rate on random operands must lie between half of the
    nominal dense peak and the peak itself (it is power-capped, never faster than the pipe), e4m3 about twice bf16; bad arguments fail."""
    g = torch.Generator().manual_seed(5)
    r = torch.randn(1 << 18, generator=g)
    bf = ops.mfma_peak_probe(r.to(BF).cuda(), fp8=False, seconds=0.6)
    f8 = ops.mfma_peak_probe(r.to(torch.float8_e4m3fn).view(torch.uint8).cuda(), fp8=True, seconds=0.6)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert bf["flops_per_launch"] % (cus * 8 * 64 * 16384) == 0 and f8["flops_per_launch"] % (cus * 8 * 32 * 65536) == 0
    assert 1250 < bf["tflops"] <= 2500 and 2500 < f8["tflops"] <= 5000, (bf, f8)
    assert 1.6 < f8["tflops"] / bf["tflops"] < 2.4
    with pytest.raises(RuntimeError, match="mfma_peak_probe"):
        ops.mfma_peak_probe(torch.zeros(64, dtype=BF, device="cuda"))            # fewer than 16 KiB of operands


def test_gemm_persistent_rejects_odd_k_tiles(ops):
    a, w = rnd((512, 192), 1).to(BF).cuda(), rnd((256, 192), 2).to(BF).cuda()
    with pytest.raises(RuntimeError, match="persistent"):
        ops.gemm(a, w, None, variant=3)


# ----------------------------------------------------------------------------- fp8 (e4m3) GEMM path, BASELINE config 5
F8 = torch.float8_e4m3fn


@pytest.mark.parametrize("shape", [(1, 37, 256), (2, 300, 3072), (3, 64, 15360)])
def test_quantize_rows_fp8_matches_torch(ops, shape):
    """scale = max|row| / 448 and bytes = torch's round-to-nearest-even e4m3 cast of the exactly rounded quotient x / scale,
    bit for bit (bf16 data with a scale of few mantissa bits sits on e4m3 rounding ties all the time, so the kernel divides
    rather than multiplying by a reciprocal)."""
    B, R, K = shape
    x = (rnd(shape, 31) * (rnd((B, R, 1), 32).abs() * 3 + 0.01)).to(BF)
    x[0, 0] = 0                                            # all-zero row -> scale 1, zeros
    q, sc = ops.quantize_rows_fp8(x.cuda())
    s_ref = x.float().abs().amax(-1) / 448.0
    s_ref = torch.where(s_ref > 0, s_ref, torch.ones_like(s_ref))
    assert torch.equal(sc.cpu(), s_ref)
    ref = (x.float() / s_ref[..., None]).clamp(-448, 448).to(F8).float()
    got = q.cpu().view(F8).float()
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref)
    assert got[0, 0].abs().max().item() == 0 and sc[0, 0].item() == 1.0


@pytest.mark.parametrize("B,M,N,K", [(1, 512, 512, 256), (2, 1000, 3136, 512), (1, 300, 264, 12288), (2, 4736, 3072, 3072)])
def test_gemm_fp8_matches_dequantised_matmul(ops, B, M, N, K):
    """tfx_gemm_fp8 against (q_a . q_w^T) * s_a * s_w + bias evaluated in fp32 on the SAME e4m3 operands, all epilogues;
    tolerance = the bf16 output rounding (max-norm 1e-2, MAE 2e-3 relative, as for the bf16 GEMM)."""
    a = (rnd((B, M, K), 41) * (rnd((B, M, 1), 42).abs() + 0.1)).to(BF).cuda()
    w = rnd((N, K), 43, 0.05).to(BF).cuda()
    bias, gate, res = rnd((N,), 44).to(BF).cuda(), rnd((B, N), 45).to(BF).cuda(), rnd((B, M, N), 46).to(BF).cuda()
    aq, sa = ops.quantize_rows_fp8(a)
    wq, sw = ops.quantize_rows_fp8(w)
    lin = (aq.view(F8).float() @ wq.view(F8).float().T) * sa[..., None] * sw[None, None, :] + bias.float()
    gf = max(0, (N // 256 - 1) * 256)
    gelu_ref = torch.cat([lin[..., :gf], torch.nn.functional.gelu(lin[..., gf:], approximate="tanh")], -1)
    cases = [(ops.EPI_BIAS, {}, lin), (ops.EPI_BIAS_GELU, dict(gelu_from_col=gf), gelu_ref),
             (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res), res.float() + (gate.float()[:, None] * lin.to(BF).float()).to(BF).float()),
             (ops.EPI_BIAS_RES, dict(res=res), res.float() + lin.to(BF).float())]
    ws = torch.empty(4 * B * M * N, dtype=torch.float32, device="cuda")   # few-tile shapes then also run split-K
    for epi, kw, ref in cases:
        close(ops.gemm_fp8(aq, sa, wq, sw, bias, epilogue=epi, **kw), ref.to(BF))
        close(ops.gemm_fp8(aq, sa, wq, sw, bias, epilogue=epi, workspace=ws, **kw), ref.to(BF))


@pytest.mark.parametrize("D", [256, 3072])
def test_ln_modulate_fp8_is_ln_modulate_then_quantize(ops, D):
    """The fp8 mode's fused LayerNorm -> e4m3 operand: bit-identical to the two separate kernels."""
    B, R = 2, 77
    x, shift, scale = rnd((B, R, D), 61).to(BF).cuda(), rnd((B, D), 62, 0.5).to(BF).cuda(), rnd((B, D), 63, 0.5).to(BF).cuda()
    x[1, 5] = 0
    shift[1].zero_()                     # -> an all-zero output row in batch 1: scale 1, zero codes
    q_ref, s_ref = ops.quantize_rows_fp8(ops.ln_modulate(x, shift, scale))
    q, s = ops.ln_modulate_fp8(x, shift, scale)
    assert torch.equal(s, s_ref) and torch.equal(q, q_ref)
    assert s[1, 5].item() == 1.0 and q[1, 5].abs().max().item() == 0


def test_gemm_fp8_rejects_unsupported_k(ops):
    aq = torch.zeros(64, 384, dtype=torch.uint8, device="cuda")
    wq = torch.zeros(64, 384, dtype=torch.uint8, device="cuda")
    one = torch.ones(64, dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="gemm_fp8"):
        ops.gemm_fp8(aq, one, wq, one)


# ----------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,N", [(1, 1, 1), (2, 3, 8), (1, 2, 33), (1, 1, 64), (1, 1, 65), (2, 2, 96), (1, 3, 300), (2, 2, 1664), (1, 24, 520)])
def test_attention(ops, B, H, N):
    q, k, v = (rnd((B, N, H * 128), s).to(BF) for s in (20, 21, 22))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * 128)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda())
    close(got, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


@pytest.mark.parametrize("N", [64, 192, 257, 1000, 4608])
def test_attention_default_kernel_is_deterministic_and_nan_free(ops, N):
    """The default kernel's schedule keeps every read of an MFMA result a fixed number of MFMAs behind its producer (hipcc
    cannot insert that wait in front of inline-asm VALU code); a violation shows up as sporadic stale rows, different from
    run to run.  Six back-to-back runs, interleaved with a kernel that dirties the register file and LDS differently, must
    be bit-identical and match fp32 SDPA."""
    B, H = 1, 3
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, N, H * 128, generator=g).to(BF).cuda() for _ in range(3))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, H * 128)
    outs = []
    for i in range(6):
        outs.append(ops.attention(q, k, v))
        if i % 2:
            torch.nn.functional.scaled_dot_product_attention(qh, kh, vh)      # another kernel's leftovers in between
    for o in outs:
        assert torch.isfinite(o.float()).all()
        assert torch.equal(o, outs[0])
    close(outs[0], ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


def test_attention_unaligned_output_rows_fail_loudly(ops):
    """The product library carries the one-wave-per-SIMD kernel only (round 6: its predecessors live in the bench library); it stores whole
    rows in 16-byte pieces, so an output view whose row stride is not a multiple of 8 elements is refused with a message -- never served
    by a silent slow path (the DiT's layouts are always aligned)."""
    B, H, N = 1, 2, 300
    q, k, v = (rnd((B, N, H * 128), s).to(BF).cuda() for s in (31, 32, 33))
    buf = torch.zeros(B, N, H * 128 + 4, dtype=BF, device="cuda")
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        ops.attention(q, k, v, out=buf[:, :, :H * 128])
    assert (buf == 0).all()
    with pytest.raises(RuntimeError, match="bench-only"):
        ops.set_option("attention_waves", 10)
    with pytest.raises(RuntimeError, match="bench-only"):
        ops.set_option("gemm_waves", 4)


@pytest.mark.parametrize("nw", [30])
def test_attention_reference_maximum_paths_vs_fp64(ops, nw):
    """Inputs that drive every path of the (lazy) reference-maximum logic, against an fp64 softmax: scores that are all
    very negative (first tile must pin the reference to the true maximum: no underflow of the row sum), a maximum that
    grows in every tile, isolated spikes in late tiles (everything accumulated so far is rescaled exactly once), and a
    peaked distribution.  Same bound for the exact-online-max kernel (8), the matrix-pipe kernels (10, and the default 30 whose
    two q-blocks per wave move their references independently) and the half-tile pipelined kernel (20: the reference can move
    in the middle of a pending P.V there)."""
    B, H, N = 2, 2, 1216
    g = torch.Generator().manual_seed(77)
    q, k, v = (torch.randn(B, N, H * 128, generator=g).to(BF) for _ in range(3))

    def ref64(q, k, v):
        qh, kh, vh = (t.double().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
        return (torch.softmax((qh @ kh.transpose(-1, -2)) * 128 ** -0.5, -1) @ vh).transpose(1, 2).reshape(B, N, H * 128)

    k_spike = k.clone()
    k_spike[0, 900, :128] = q[0, 17, :128] * 6.0
    k_spike[1, 1100, 128:] = q[1, 300, 128:] * 9.0
    ramp = (torch.arange(N).view(1, N, 1) / N * 6).to(BF).expand(B, N, H * 128).contiguous()
    # option 32 keeps a row's reference at 0 while its scores stay inside +-64 (exp2 domain) and issues the reference MFMA only
    # while some row of the wave has one: "far negative" (every score around -110: the first tile must pin), "wide" (row maxima on
    # both sides of 64 inside one wave: pinned and unpinned rows side by side from the first tile on; planted keys rather than scaled
    # queries: at |score| ~ 100 the one extra bf16 rounding of the pre-scaled q -- common to every kernel but the textbook one -- moves
    # near-tied weights by more than the bound, which is a property of the scaling, not of the reference logic under test) and "late far spike" (a +150 score in a late tile of
    # a row whose reference was 0 until then) drive those paths; the other kernels must of course get them right too
    k_wide = k.clone()                   # sixteen keys of the FIRST tile each aligned with one even query row of the first wave: those rows
    for j in range(16):                  # see a score around +114 there (their neighbours at most ~ +-30 from the same keys)
        k_wide[0, j, :128] = q[0, 2 * j, :128] * 7.0
        k_wide[1, 32 + j, 128:] = q[1, 64 + 2 * j, 128:] * 7.0
    k_far = k.clone()
    k_far[0, 1000, :128] = q[0, 40, :128] * 9.0
    k_far[1, 70, 128:] = q[1, 1200, 128:] * 10.0
    cases = {"very negative": (q.abs() + 0.5, -(k.abs() + 0.5) * 2, v), "growing maximum": (q.abs() + 0.1, ramp, v),
             "late spikes": (q, k_spike, v), "peaked": (q * 3, k, v),
             "far negative": (q.abs() + 0.5, -(k.abs() + 0.5) * 4, v), "wide": (q, k_wide, v), "late far spike": (q * 0.5, k_far, v)}
    ops.set_option("attention_waves", nw)
    try:
        for name, (qq, kk, vv) in cases.items():
            got = ops.attention(qq.cuda(), kk.cuda(), vv.cuda()).double().cpu()
            ref = ref64(qq, kk, vv)
            assert torch.isfinite(got).all(), name
            err = (got - ref).abs()
            assert err.max().item() <= 2e-2 * ref.abs().max().item() + 1e-3, (name, err.max().item())
            assert err.mean().item() <= 6e-3 * ref.abs().mean().item() + 1e-5, (name, err.mean().item(), ref.abs().mean().item())
    finally:
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)


def test_attention_score_bound_selects_the_reference_free_stream(ops):
    """tfx_attn_args.score_bound (the caller's promise |scale q.k| <= bound; the DiT derives it from the q / k RMSNorm weights):
    with a bound of at most 41 the default kernel runs without any reference maximum (attn_w4_kernel<4>, option 34) -- same
    softmax, checked against fp32 SDPA; a bound that is too large, or none, leaves the guarded kernel in place (bit-identical
    to a call without the field); `attention_use_bound 0` switches the shortcut off."""
    B, H, N = 2, 3, 1500
    q, k, v = (rnd((B, N, H * 128), s).to(BF) for s in (41, 42, 43))
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    sc = (qh @ kh.transpose(-1, -2)) * 128 ** -0.5
    bound = sc.abs().max().item() * 1.01
    assert bound < 41
    ref = torch.softmax(sc, -1) @ vh
    ref = ref.transpose(1, 2).reshape(B, N, H * 128)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    plain = ops.attention(qc, kc, vc)
    bounded = ops.attention(qc, kc, vc, score_bound=bound)
    close(bounded, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    assert (bounded.float() - plain.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert torch.equal(ops.attention(qc, kc, vc, score_bound=100.0), plain)          # bound too large: the guarded kernel
    try:
        ops.set_option("attention_waves", 34)
        assert torch.equal(ops.attention(qc, kc, vc, score_bound=bound), bounded)    # 34 with a bound = what the default picked
        close(ops.attention(qc, kc, vc), ref.to(BF), max_rel=2e-2, mae_rel=4e-3)     # 34 without one: the guarded kernel (product library; the bench library's lazy-reference form 33 is tested in tools/variant_tests/)
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)
        ops.set_option("attention_use_bound", 0)
        assert torch.equal(ops.attention(qc, kc, vc, score_bound=bound), plain)      # shortcut off: the field is ignored
    finally:
        ops.set_option("attention_waves", ops.DEFAULT_ATTENTION)
        ops.set_option("attention_use_bound", 1)
    # ragged N, tiny N and the first-tile-only case through the reference-free stream
    for (b, h, n) in [(1, 1, 1), (2, 3, 8), (1, 2, 33), (1, 1, 65), (1, 3, 300)]:
        q2, k2, v2 = (rnd((b, n, h * 128), s).to(BF) for s in (51, 52, 53))
        q2h, k2h, v2h = (t.float().view(b, n, h, 128).transpose(1, 2) for t in (q2, k2, v2))
        r2 = torch.nn.functional.scaled_dot_product_attention(q2h, k2h, v2h).transpose(1, 2).reshape(b, n, h * 128)
        close(ops.attention(q2.cuda(), k2.cuda(), v2.cuda(), score_bound=30.0), r2.to(BF), max_rel=2e-2, mae_rel=4e-3)


@pytest.mark.parametrize("B,H,N", [(2, 24, 2000), (16, 24, 50), (3, 24, 1100), (1, 24, 4608)])
def test_attention_persistent_form_is_bit_identical_to_one_workgroup_per_item(ops, B, H, N):
    """Round 4: with more (b, h, q-tile) items than CUs the default kernel runs as ONE workgroup per CU walking its items, the next
    item's Q rows and first K / V tile requested inside the current item's last tile (attention_persistent, on by default).  Same
    arithmetic per item in the same order: bit-identical to the one-workgroup-per-item launch, for ragged N, single-tile items
    (the first tile is also the last), with and without the reference-free stream, and against fp32 SDPA on sampled heads."""
    g = torch.Generator().manual_seed(B * 1000 + N)
    q, k, v = (torch.randn(B, N, H * 128, generator=g).to(BF).cuda() for _ in range(3))
    outs = {}
    try:
        for bound in (0.0, 30.0):
            for pers in (1, 0):
                ops.set_option("attention_persistent", pers)
                outs[(bound, pers)] = ops.attention(q, k, v, score_bound=bound)
            assert torch.equal(outs[(bound, 1)], outs[(bound, 0)]), bound
            assert torch.equal(ops.attention(q, k, v, score_bound=bound), outs[(bound, 0)])      # and again (determinism)
    finally:
        ops.set_option("attention_persistent", 1)
    for (bi, hi) in ((0, 0), (B - 1, H - 1), (B // 2, 7)):
        sl = slice(hi * 128, (hi + 1) * 128)
        ref = torch.nn.functional.scaled_dot_product_attention(q[bi:bi + 1, :, sl].float()[:, None], k[bi:bi + 1, :, sl].float()[:, None],
                                                               v[bi:bi + 1, :, sl].float()[:, None])[:, 0]
        for o in outs.values():
            close(o[bi:bi + 1, :, sl], ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


def test_attention_online_softmax_rescale_branch(ops):
    """A key far above the rest in a LATE tile forces the running-max rescale of the accumulated output."""
    B, H, N = 1, 1, 256
    q, k, v = (rnd((B, N, 128), s).to(BF) for s in (23, 24, 25))
    k[0, 200] = q[0, 17] * 4.0  # q17 . k200 >> anything seen in tiles 0..2
    ref = torch.nn.functional.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None])[:, 0]
    close(ops.attention(q.cuda(), k.cuda(), v.cuda()), ref.to(BF), max_rel=2e-2, mae_rel=4e-3)


@pytest.mark.parametrize("B,N", [(1, 3100), (2, 4608)])
def test_attention_tail_split_matches_unsplit_and_reference(ops, B, N):
    """When the last round of workgroups is partly filled, the default kernel cuts its q-tiles into key ranges and a merge kernel
    finishes them (attention_w4.hip, option attention_tail_split = 1, off by default: 312 / 864 workgroups on 256 CUs here, two
    ranges of >= 24 key tiles, a ragged last tile at N = 3100).  The split result must agree with the unsplit kernel to bf16 rounding, with the fp32 reference like every other
    shape, be deterministic, and survive the in-place-over-q layout of the blocks."""
    H, D = 24, 24 * 128
    y = (rnd((B, N, 4 * D), 61) * 1.5).to(BF).cuda()
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:3 * D]
    qh, kh, vh = (t.float().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, N, D)     # fp32, on the GPU
    try:
        ops.set_option("attention_tail_split", 0)
        plain = ops.attention(q, k, v)
        ops.set_option("attention_tail_split", 1)
        split = ops.attention(q, k, v)
        again = ops.attention(q, k, v)
        pad_before = y[:, :, 3 * D:].clone()
        ops.attention(q, k, v, out=q)
        inplace_q, pad_after = y[:, :, 2 * D:3 * D].clone(), y[:, :, 3 * D:].clone()
    finally:
        ops.set_option("attention_tail_split", 0)     # the default: off (batch-size invariance, see attention_w4.hip)
    assert torch.equal(split, again)
    assert not torch.equal(split, plain)                        # the split path really ran (different summation order somewhere)
    close(split, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    close(plain, ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    d = (split.float() - plain.float()).abs()
    assert d.max().item() <= 2.0 ** -7 * ref.abs().max().item() and d.mean().item() <= 1e-3 * ref.abs().mean().item()
    assert torch.equal(inplace_q, split) and torch.equal(pad_after, pad_before)
    # the partials are the only device memory the library allocates itself: the release hook frees them (138 MB per stream that used the
    # knob), the next launch with the knob on allocates afresh and computes the same bits
    free0 = torch.cuda.mem_get_info()[0]
    ops.release_scratch()
    assert torch.cuda.mem_get_info()[0] >= free0 + (100 << 20)
    try:
        ops.set_option("attention_tail_split", 1)
        y2 = (rnd((B, N, 4 * D), 61) * 1.5).to(BF).cuda()          # q was overwritten in place above: the same inputs again
        assert torch.equal(ops.attention(y2[:, :, 2 * D:3 * D], y2[:, :, :D], y2[:, :, D:2 * D]), split)
    finally:
        ops.set_option("attention_tail_split", 0)
        ops.release_scratch()


@pytest.mark.parametrize("B,H,N,bound", [(1, 24, 4608, 30.0), (8, 24, 1100, 30.0), (2, 5, 2304, 0.0), (3, 24, 2304, 30.0), (1, 24, 1664, 0.0),
                                         (4, 24, 3100, 30.0), (8, 24, 4608, 30.0), (5, 24, 4608, 30.0), (1, 3, 8704, 30.0), (2, 24, 5000, 0.0),
                                         (1, 48, 1100, 30.0)])   # ... groups of 51 CUs, three heads, a ragged last key tile inside a dealt piece, 48 heads
def test_attention_streamk_dealing_matches_whole_items(ops, B, H, N, bound):
    """Round 6: with a workspace the persistent kernel deals the items of a sample's partly filled LAST round as (item, 64-key tile) units to
    the sample's group of CUs instead of leaving part of the chip idle (attention_w4.hip::w4_sk_bound; option attention_streamk: 1 = when
    the estimate says it pays, 2 = whenever admissible, 0 = never).  A tail item cut by a CU boundary leaves un-normalised partials that a
    merge pass adds: another fp32 summation order for those rows, nothing else.  Cases: the headline launch at batch 1 and 8, ragged N,
    launches with fewer items than CUs whose items are cut into up to three parts (share < nkv), B = 3 (a group of 85 CUs per sample, one CU
    idle), few heads, both streams (bound 0 = the guarded kernel).  Asserted: close to the
    unsplit kernel and to the fp32 reference, deterministic, identical samples of a batch identical, in-place over q, and -- without a
    workspace or with the option off -- bit-identical to the whole-item form."""
    D = H * 128
    g = torch.Generator().manual_seed(1000 + N + B)
    one = (torch.randn(1, N, 4 * D, generator=g) * 1.2).to(BF)
    y = torch.cat([one] + [(torch.randn(1, N, 4 * D, generator=g) * 1.2).to(BF) for _ in range(B - 2)] + ([one] if B > 1 else []), 0).cuda()   # first == last sample
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:3 * D]
    ws = torch.empty(72 << 20, dtype=torch.uint8, device="cuda")
    bi = 0
    qh, kh, vh = (t[bi:bi + 1].float().view(1, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(1, N, D)
    try:
        ops.set_option("attention_streamk", 0)
        plain = ops.attention(q, k, v, score_bound=bound, workspace=ws)
        ops.set_option("attention_streamk", 2)
        no_ws = ops.attention(q, k, v, score_bound=bound)
        ops.attention_mode_counts(reset=True)
        dealt = ops.attention(q, k, v, score_bound=bound, workspace=ws)
        assert ops.attention_mode_counts()["streamk_tail"] == 1
        again = ops.attention(q, k, v, score_bound=bound, workspace=ws)
        pad_before = y[:, :, 3 * D:].clone()
        ops.attention(q, k, v, out=q, score_bound=bound, workspace=ws)
        inplace_q, pad_after = y[:, :, 2 * D:3 * D].clone(), y[:, :, 3 * D:].clone()
    finally:
        ops.set_option("attention_streamk", 1)
    assert torch.equal(no_ws, plain)
    assert torch.equal(dealt, again) and torch.isfinite(dealt.float()).all()
    assert not torch.equal(dealt, plain)                        # some item was cut: the dealt path really ran
    if B > 1:
        assert torch.equal(dealt[0], dealt[B - 1])              # every sample is dealt to its own group of CUs by the same rule
    close(dealt[bi:bi + 1], ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    d = (dealt.float() - plain.float()).abs()
    # (guarded kernel, bound 0: every part rounds its bf16 weights against its OWN reference maximum -- more than a summation order)
    assert d.max().item() <= 2.0 ** -7 * ref.abs().max().item() and d.mean().item() <= (1e-3 if bound > 0 else 3e-3) * ref.abs().mean().item()
    frac = (d > 0).float().mean().item()
    print(f"B {B} H {H} N {N} bound {bound}: {frac:.4f} of the outputs differ from the whole-item form, max {d.max().item():.3g}")
    assert frac < 0.6
    assert torch.equal(inplace_q, dealt) and torch.equal(pad_after, pad_before)


def test_attention_strided_inplace_over_q(ops):
    """Layout used by the blocks: [k | v | q | pad] rows, output written over q."""
    B, N, H = 2, 200, 2
    D = H * 128
    y = rnd((B, N, 4 * D), 26).to(BF).cuda()
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:3 * D]
    qc, kc, vc = (t.float().cpu().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qc, kc, vc).transpose(1, 2).reshape(B, N, D)
    pad_before = y[:, :, 3 * D:].clone()
    ops.attention(q, k, v, out=q)
    close(y[:, :, 2 * D:3 * D], ref.to(BF), max_rel=2e-2, mae_rel=4e-3)
    assert torch.equal(y[:, :, 3 * D:], pad_before)


# ----------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("D", [256, 3072])
def test_ln_modulate_matches_bf16_oracle(ops, D):
    B, R = 2, 37
    x, shift, scale = rnd((B, R, D), 30).to(BF), rnd((B, D), 31, 0.5).to(BF), rnd((B, D), 32, 0.5).to(BF)
    ref = fo.layer_norm(x) * (1 + scale[:, None]) + shift[:, None]  # bf16 op chain of the reference
    got = ops.ln_modulate(x.cuda(), shift.cuda(), scale.cuda()).cpu()
    # identical rounding chain; the only freedom is the fp32 summation order of mean/var -> allow 1 bf16 ulp rarely
    diff = (got.float() - ref.float()).abs()
    ulp = ref.float().abs() * 2 ** -7 + 4e-3   # 1 bf16 ulp of the value (+ 1 ulp of an O(0.5) intermediate near zero)
    assert (diff <= ulp).all()
    assert (diff > 0).float().mean().item() < 0.02


def test_rmsnorm_rope_matches_bf16_oracle(ops, golden):
    g = golden("g1_ops")
    B, H, N, T = 2, 3, 29, 5
    x = g["rope.x"].to(BF)                       # [2,3,29,128] heads-major as the reference sees it
    wq, wk, waq, wak = (1 + 0.1 * rnd((128,), 40)).to(BF), (1 + 0.1 * rnd((128,), 41)).to(BF), \
        (1 + 0.1 * rnd((128,), 42)).to(BF), (1 + 0.1 * rnd((128,), 43)).to(BF)
    cos, sin = g["rope.cos"], g["rope.sin"]

    def ref(xh, w_txt, w_img):
        y = torch.cat([fo.rms_norm(xh[:, :, :T], w_txt), fo.rms_norm(xh[:, :, T:], w_img)], 2)
        return fo.apply_rope(y, cos, sin)

    k = rnd((2, 3, 29, 128), 44).to(BF)
    buf = torch.zeros(B, N, 3 * H * 128, dtype=BF)
    buf[:, :, :H * 128] = k.transpose(1, 2).reshape(B, N, -1)
    buf[:, :, 2 * H * 128:] = x.transpose(1, 2).reshape(B, N, -1)
    got = ops.rmsnorm_rope_(buf.cuda(), 2 * H * 128, 0, H, T, wq.cuda(), wk.cuda(), waq.cuda(), wak.cuda(),
                            cos.cuda(), sin.cuda()).cpu()
    rq = ref(x, waq, wq).transpose(1, 2).reshape(B, N, -1)
    rk = ref(k, wak, wk).transpose(1, 2).reshape(B, N, -1)
    for gt, rf in ((got[:, :, 2 * H * 128:], rq), (got[:, :, :H * 128], rk)):
        diff = (gt.float() - rf.float()).abs()
        assert (diff <= rf.float().abs() * 2 ** -7 + 4e-3).all()
        assert (diff > 0).float().mean().item() < 0.02
    assert got[:, :, H * 128:2 * H * 128].abs().max().item() == 0  # v columns untouched


def test_scheduler_steps_bitexact_vs_reference_trajectories(ops, golden):
    g = golden("g4_sched")
    n, S = 6, 4096
    mu = so.calculate_shift(S, 256, 4096, 0.5, 1.15)
    lin = so.pipeline_sigmas(n)
    es, as_ = so.euler_sigmas(lin, mu), so.amo_sigmas(lin, mu)
    # the reference multiplies the bf16 model output by a 0-dim f32 tensor: TensorIterator casts that scalar to the
    # common dtype (bf16) first, so the Euler dsigma and the AMO (t_over - t) are bf16-rounded (a, b stay f32)
    ecoef = (es[1:] - es[:-1]).to(BF).float().cuda()
    acoef = torch.tensor([so.amo_coefficients(as_[i].item(), as_[i + 1].item(), 2.0) for i in range(n)],
                         dtype=torch.float32)
    acoef[:, 0] = acoef[:, 0].to(BF).float()
    acoef = acoef.cuda()
    x = g["traj.x0"].to(BF).cuda()
    xin = torch.zeros(2, 16, 96, dtype=BF, device="cuda")
    for i in range(n):
        ops.euler_step_(g[f"traj.v{i}"].to(BF).cuda(), x, ecoef, step=i, xin=xin)
        assert torch.equal(x.cpu(), g[f"traj.euler.bf16.x{i}"])
        assert torch.equal(xin[:, :, :64], x) and xin[:, :, 64:].abs().max().item() == 0
    x = g["traj.x0"].to(BF).cuda()
    step_ptr = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in range(n):
        ops.amo_step_(g[f"traj.v{i}"].to(BF).cuda(), x, acoef, g[f"traj.amo.eps{i}"].cuda(), step_ptr=step_ptr)
        ops.advance_step_(step_ptr)
        assert torch.equal(x.cpu(), g[f"traj.amo.bf16.x{i}"])
    assert step_ptr.item() == n


def test_timestep_embedding(ops, golden):
    g = golden("g1_ops")
    got = ops.timestep_embedding(g["tsemb.t"].cuda()).cpu().float()
    ref = g["tsemb.out"].to(BF).float()
    assert (got - ref).abs().max().item() <= 2 ** -7  # values in [-1,1]: 1 bf16 ulp (device sinf/cosf vs torch CPU)


def test_small_ops(ops):
    a, b = rnd((3, 520), 50).to(BF), rnd((3, 520), 51).to(BF)
    assert torch.equal(ops.add(a.cuda(), b.cuda()).cpu(), a + b)
    s = ops.silu(a.cuda()).cpu()
    assert (s.float() - torch.nn.functional.silu(a).float()).abs().max().item() <= 2 ** -6
    src = rnd((2, 5, 64), 52).to(BF).cuda()
    dst = torch.zeros(2, 5, 384, dtype=BF, device="cuda")
    ops.scatter_cols_(src, dst, 64)
    assert torch.equal(dst[:, :, 64:128], src) and dst[:, :, :64].abs().max().item() == 0
    big = rnd((2, 9, 256), 53).to(BF).cuda()
    out = torch.zeros(2, 20, 256, dtype=BF, device="cuda")
    ops.copy_rows_(big, out[:, 4:13])
    assert torch.equal(out[:, 4:13], big) and out[:, :4].abs().max().item() == 0


# ----------------------------------------------------------------------------- round 5: the boundary's remaining names
def test_gate_residual_is_the_references_two_bf16_ops_bit_for_bit(ops):
    """tfx_gate_residual: `hidden_states + gate.unsqueeze(1) * attn_output` (transformer_flux.py:733-735, 817-818) -- two bf16 ops in the
    reference, reproduced bit for bit (ragged rows, strided views of wider buffers, in place over the residual), and equal to the
    GEMM's gated-residual epilogue on the same rounded Linear output."""
    B, R, D = 3, 37, 3072
    x, g, r = rnd((B, R, D), 60).to(BF), rnd((B, D), 61).to(BF), rnd((B, R, D), 62).to(BF)
    ref = r + g.unsqueeze(1) * x                                     # torch bf16: product rounded, then the add rounded
    assert torch.equal(ops.gate_residual(x.cuda(), g.cuda(), r.cuda()).cpu(), ref)
    wide = torch.zeros(B, R + 3, 2 * D, dtype=BF, device="cuda")     # x as a column slice, the result written in place over res
    wide[:, 2:2 + R, D:] = x.cuda()
    res = r.cuda().clone()
    out = ops.gate_residual(wide[:, 2:2 + R, D:], g.cuda(), res, out=res)
    assert out.data_ptr() == res.data_ptr() and torch.equal(res.cpu(), ref)
    a, w, bias = rnd((B, R, 256), 63).to(BF), rnd((D, 256), 64, 0.05).to(BF), rnd((D,), 65).to(BF)
    y = ops.gemm(a.cuda(), w.cuda(), bias.cuda())                    # the Linear's bf16 output, then the standalone op ...
    fused = ops.gemm(a.cuda(), w.cuda(), bias.cuda(), epilogue=2, gate=g.cuda(), res=r.cuda())      # ... == the fused epilogue
    assert torch.equal(ops.gate_residual(y, g.cuda(), r.cuda()), fused)


def test_rmsnorm_rope_qk_is_the_same_entry_point_under_survey_8b_name(ops, golden):
    from textflux_amd import _lib as L
    g = golden("g1_ops")
    B, H, N, T = 2, 3, 29, 5
    buf = rnd((B, N, 3 * H * 128), 70).to(BF).cuda()
    w = [(1 + 0.1 * rnd((128,), 71 + i)).to(BF).cuda() for i in range(4)]
    cos, sin = g["rope.cos"].cuda().contiguous(), g["rope.sin"].cuda().contiguous()
    a, b = buf.clone(), buf.clone()
    ops.rmsnorm_rope_(a, 2 * H * 128, 0, H, T, *w, cos, sin)
    L.check(L.lib().tfx_rmsnorm_rope_qk(b.data_ptr(), b.stride(1), b.stride(0), 2 * H * 128, 0, H, N, T, B, w[0].data_ptr(),
                                        w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(), cos.data_ptr(), sin.data_ptr(), 1e-6,
                                        torch.cuda.current_stream().cuda_stream), "rmsnorm_rope_qk")
    assert torch.equal(a, b) and not torch.equal(a, buf)


@pytest.mark.parametrize("N", [2304, 4608])      # 4608 = the P1024 joint sequence, the largest N the headline launches
def test_attention_reference_free_stream_holds_up_to_the_edge_of_its_admissible_range(ops, N):
    """The reference-free stream is admissible while score_bound * log2 e + log2 N + 24 <= 126 (attention.hip): scores of BOTH signs at
    ~97 % of that bound in the same launch -- rows whose every score is near -b (weights 2^-84, the sum must not underflow), rows with
    every score near +b (2^+84 summed over N keys), rows that mix both (the small weights vanish, as in exact arithmetic) -- with
    |v| up to 4096, against fp64 softmax; the counters show which stream ran; one step beyond the bound the guarded kernel takes over."""
    B, H = 1, 2
    lim = (126.0 - 24.0 - torch.log2(torch.tensor(float(N))).item()) / 1.4426950408889634          # 62.9 at N = 2304, 62.2 at 4608
    u = torch.nn.functional.normalize(rnd((128,), 80), dim=0)
    amp = (0.97 * lim / 128 ** -0.5) ** 0.5
    g = torch.Generator().manual_seed(81)
    sq = torch.where(torch.rand(B, N, H, 1, generator=g) < 0.5, -1.0, 1.0)       # sign of every query / key along u
    sk = torch.where(torch.rand(B, N, H, 1, generator=g) < 0.5, -1.0, 1.0)
    sk[:, :, 1] = 1.0                                                            # head 1: every key +u -> whole rows at +b or at -b
    q = (sq * amp * u + 0.05 * torch.randn(B, N, H, 128, generator=g)).reshape(B, N, H * 128).to(BF)
    k = (sk * amp * u + 0.05 * torch.randn(B, N, H, 128, generator=g)).reshape(B, N, H * 128).to(BF)
    v = (torch.randn(B, N, H * 128, generator=g) * 1024).clamp(-4096, 4096).to(BF)
    qh, kh, vh = (t.double().view(B, N, H, 128).transpose(1, 2) for t in (q, k, v))
    sc = (qh @ kh.transpose(-1, -2)) * 128 ** -0.5
    bound = sc.abs().max().item() * 1.001
    assert 0.9 * lim < bound < lim and sc.min().item() < -0.9 * lim and sc.max().item() > 0.9 * lim
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, N, H * 128)
    ops.attention_mode_counts(reset=True)
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), score_bound=bound)
    assert ops.attention_mode_counts()["w4_reference_free"] == 1
    close(got, ref.to(BF), max_rel=2e-2, mae_rel=6e-3)
    guarded = ops.attention(q.cuda(), k.cuda(), v.cuda())
    close(guarded, ref.to(BF), max_rel=2e-2, mae_rel=6e-3)
    ops.attention_mode_counts(reset=True)
    assert torch.equal(ops.attention(q.cuda(), k.cuda(), v.cuda(), score_bound=lim * 1.02), guarded)     # beyond the range: the guard stays
    c = ops.attention_mode_counts(reset=True)
    assert c["w4_guarded"] == 1 and c["w4_reference_free"] == 0
