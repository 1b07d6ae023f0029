"""Out-of-bounds canaries around the output of every kernel of the DiT step ("race detection / sanitizers": the
reference has none; the build owns it).  Every kernel's output is a VIEW inside a larger buffer pre-filled with a canary pattern -- rows
before and after the view, columns beyond the view's width (row stride > row length), the batch gap -- at ragged sizes that exercise
the edge tiles (the persistent GEMM drops out-of-range stores through its buffer descriptors' range check; the attention kernel pads
query rows and requests key tiles past the end); after the launch every canary byte must be untouched and the view must be fully
written.  Also covers the in-place forms (gate residual over res, attention over q)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
CANARY = -2.0 ** 127            # bf16 0xFF00: a value no kernel here produces


@pytest.fixture(scope="module")
def ops():
    from textflux_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    return (torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale).to(BF).cuda()


def framed(B, R, C, pad_rows=3, pad_cols=16):
    """A [B, R, C] view inside a canary-filled [B, R + 2 pad_rows, C + pad_cols] buffer."""
    buf = torch.full((B, R + 2 * pad_rows, C + pad_cols), CANARY, dtype=BF, device="cuda")
    return buf, buf[:, pad_rows:pad_rows + R, :C]


def check(buf, view, what):
    mask = torch.ones_like(buf, dtype=torch.bool)
    pr = (buf.shape[1] - view.shape[1]) // 2
    mask[:, pr:pr + view.shape[1], :view.shape[2]] = False
    assert (buf[mask] == CANARY).all(), f"{what}: wrote outside its output"
    assert (view != CANARY).all() and torch.isfinite(view.float()).all(), f"{what}: left part of its output unwritten"


@pytest.mark.parametrize("B,M,N,K", [(2, 300, 264, 128), (1, 1000, 3136, 256), (3, 37, 72, 192), (2, 513, 520, 384)])
def test_gemm_every_epilogue_stays_inside_its_output(ops, B, M, N, K):
    a, w, bias = rnd((B, M, K), 1), rnd((N, K), 2, 0.05), rnd((N,), 3)
    gate, res = rnd((B, N), 4), rnd((B, M, N), 5)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for epi, kw in ((ops.EPI_BIAS, {}), (ops.EPI_BIAS_GELU, dict(gelu_from_col=0)), (ops.EPI_BIAS_GATE_RES, dict(gate=gate, res=res)),
                    (ops.EPI_BIAS_RES, dict(res=res))):
        for variant, extra in ((0, {}), (1, {}), (2, {}), (1, dict(workspace=ws))):      # generic, persistent, one-tile, persistent + K slices
            buf, out = framed(B, M, N)
            ops.gemm(a, w, bias, out=out, epilogue=epi, variant=variant, **kw, **extra)
            check(buf, out, f"gemm epi {epi} variant {variant} ws {bool(extra)}")
    # in place over the residual (how the blocks use it), the residual itself a framed view
    buf, r = framed(B, M, N)
    r.copy_(res)
    ops.gemm(a, w, bias, out=r, epilogue=ops.EPI_BIAS_GATE_RES, gate=gate, res=r)
    check(buf, r, "gemm gated residual in place")


def test_elementwise_kernels_stay_inside_their_outputs(ops):
    B, R, D = 2, 37, 3072
    x, sh, sc = rnd((B, R, D), 10), rnd((B, D), 11, 0.1), rnd((B, D), 12, 0.1)
    buf, out = framed(B, R, D)
    ops.ln_modulate(x, sh, sc, out=out)
    check(buf, out, "ln_modulate")
    buf, out = framed(B, R, D)
    ops.gate_residual(x, sh, rnd((B, R, D), 13), out=out)
    check(buf, out, "gate_residual")
    # rmsnorm + rope in place on the q / k column ranges of a [k | v | q] buffer: v and the frame stay as they were
    H, N, T = 3, 29, 5
    buf, y = framed(B, N, 3 * H * 128)
    y.copy_(rnd((B, N, 3 * H * 128), 14))
    v_before = y[:, :, H * 128:2 * H * 128].clone()
    w4 = [(1 + 0.1 * torch.randn(128, generator=torch.Generator().manual_seed(20 + i))).to(BF).cuda() for i in range(4)]
    cs = torch.randn(N, 128, generator=torch.Generator().manual_seed(30)).cuda()
    ops.rmsnorm_rope_(y, 2 * H * 128, 0, H, T, *w4, cs.contiguous(), cs.contiguous())
    check(buf, y, "rmsnorm_rope")
    assert torch.equal(y[:, :, H * 128:2 * H * 128], v_before)
    # Euler step: the new latents also go into columns 0 .. 63 of the x_embedder input [B, S, 384]; its other columns are not its business
    xin = torch.full((1, 40, 384), CANARY, dtype=BF, device="cuda")
    lat, v = rnd((1, 40, 64), 15), rnd((1, 40, 64), 16)
    coef = torch.tensor([-0.03125, -0.0625], dtype=torch.float32, device="cuda")
    ops.euler_step_(v, lat, coef, step=1, xin=xin)
    assert torch.equal(xin[:, :, :64], lat) and (xin[:, :, 64:] == CANARY).all()


@pytest.mark.parametrize("B,H,N", [(2, 3, 300), (1, 2, 65), (1, 24, 1664)])
def test_attention_stays_inside_its_output_also_in_place_over_q(ops, B, H, N):
    D = H * 128
    buf, y = framed(B, N, 3 * D, pad_rows=5, pad_cols=64)           # [k | v | q] rows as the blocks lay them out
    y.copy_(rnd((B, N, 3 * D), 40))
    k, v, q = y[:, :, :D], y[:, :, D:2 * D], y[:, :, 2 * D:]
    kv_before = y[:, :, :2 * D].clone()
    obuf, o = framed(B, N, D, pad_rows=5, pad_cols=64)
    for bound in (0.0, 30.0):
        o.fill_(CANARY)
        ops.attention(q, k, v, out=o, score_bound=bound)
        check(obuf, o, f"attention bound {bound}")
    ops.attention(q, k, v, out=q, score_bound=30.0)                 # the blocks' form: the output overwrites q
    check(buf, y, "attention in place over q")
    assert torch.equal(y[:, :, :2 * D], kv_before) and torch.equal(q, o)
