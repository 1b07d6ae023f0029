"""Torch-tensor front end of the C-ABI kernels (device memory + stream plumbing only, no math here).

Every function enqueues HIP kernels from libtextflux_hip.so on torch's current stream and returns torch
tensors that alias / own the outputs.  Inputs must live on a ROCm device; bf16 unless stated.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES, EPI_BIAS_RES = 0, 1, 2, 3
DEFAULT_ATTENTION = 0      # tfx_set_option("attention_waves", 0): back to the library's default kernel (30 = attn_w4_kernel)
BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("textflux_amd ops need tensors on the ROCm device (no CPU fallback exists)")


def _rows_view(t: torch.Tensor):
    """(ptr, ld, batch_stride, rows_per_batch, batch) of a [.., rows, cols] tensor with unit inner stride."""
    assert t.stride(-1) == 1
    if t.dim() == 2:
        return t.data_ptr(), t.stride(0), 0, t.shape[0], 1
    assert t.dim() == 3
    return t.data_ptr(), t.stride(1), t.stride(0), t.shape[1], t.shape[0]


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         epilogue: int = EPI_BIAS, gelu_from_col: int = 0, gate: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, variant: int = -1, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = epi(a[b] @ w.T + bias).  a: [M,K] or [B,M,K] (row/batch strided views allowed), w: [N,K].
    workspace: optional device scratch tensor; lets the auto path split K when the GEMM has fewer tiles than CUs."""
    _chk_dev(a, w, bias, out, gate, res, workspace)
    assert a.dtype == BF16 and w.dtype == BF16 and w.dim() in (2, 3) and w.stride(-1) == 1
    ap, lda, abs_, M, batch = _rows_view(a)
    N, K = w.shape[-2:]
    assert a.shape[-1] == K and (w.dim() == 2 or (a.dim() == 3 and w.shape[0] == batch)), "w [N, K], or [B, N, K] with a [B, M, K]"
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=BF16, device=a.device)
    cp, ldc, cbs, M2, b2 = _rows_view(out)
    assert (M2, b2) == (M, batch) and out.shape[-1] == N
    g = L.GemmArgs()
    g.A, g.lda, g.a_bstride = ap, lda, abs_
    g.W, g.ldw, g.bias = w.data_ptr(), w.stride(-2), _p(bias)
    g.w_bstride = w.stride(0) if w.dim() == 3 else 0
    g.C, g.ldc, g.c_bstride = cp, ldc, cbs
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.epilogue, g.gelu_from_col = epilogue, gelu_from_col
    if epilogue in (EPI_BIAS_GATE_RES, EPI_BIAS_RES):
        assert res is not None
        if epilogue == EPI_BIAS_GATE_RES:
            assert gate is not None and gate.stride(-1) == 1
            g.gate = gate.data_ptr()
            g.gate_bstride = gate.stride(0) if gate.dim() == 2 else 0
        rp, ldr, rbs, _, _ = _rows_view(res)
        g.res, g.ldr, g.r_bstride = rp, ldr, rbs
    if workspace is not None:
        g.workspace, g.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    L.check(L.lib().tfx_gemm_bf16(C.byref(g), variant, _stream()), "gemm")
    return out


def gemm_qkn(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], norm_q: torch.Tensor, norm_k: torch.Tensor,
             rope_cs: torch.Tensor, q_range, k_range, out: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS,
             gelu_from_col: int = 0, pos0: int = 0, eps: float = 1e-6, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tfx_gemm_bf16_qkn: the fused [k | v | q (| mlp)] projection of a FLUX block with the per-head RMSNorm + RoPE of the q / k column
    ranges applied in the GEMM's epilogue (what tfx_dit_forward launches).  rope_cs: fp32 [rows, 64, 2] (cos, sin) pairs."""
    _chk_dev(a, w, bias, out, norm_q, norm_k, rope_cs, workspace)
    assert a.dtype == BF16 and w.dtype == BF16 and w.dim() == 2 and w.stride(1) == 1
    assert norm_q.dtype == norm_k.dtype == BF16 and norm_q.numel() == norm_k.numel() == 128 and rope_cs.dtype == torch.float32 and rope_cs.is_contiguous()
    ap, lda, abs_, M, batch = _rows_view(a)
    N, K = w.shape
    assert a.shape[-1] == K and rope_cs.shape[-2:] == (64, 2) and rope_cs.shape[0] >= pos0 + M
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=BF16, device=a.device)
    cp, ldc, cbs, M2, b2 = _rows_view(out)
    assert (M2, b2) == (M, batch) and out.shape[-1] == N
    g = L.GemmArgs()
    g.A, g.lda, g.a_bstride = ap, lda, abs_
    g.W, g.ldw, g.bias = w.data_ptr(), w.stride(0), _p(bias)
    g.C, g.ldc, g.c_bstride = cp, ldc, cbs
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.epilogue, g.gelu_from_col = epilogue, gelu_from_col
    if workspace is not None:
        g.workspace, g.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    q = L.QknArgs()
    q.norm_q, q.norm_k, q.rope_cs = norm_q.data_ptr(), norm_k.data_ptr(), rope_cs.data_ptr()
    q.pos0, q.q0, q.q1, q.k0, q.k1, q.eps = pos0, q_range[0], q_range[1], k_range[0], k_range[1], eps
    L.check(L.lib().tfx_gemm_bf16_qkn(C.byref(g), C.byref(q), _stream()), "gemm_qkn")
    return out


def quantize_rows_fp8(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """Per-row absmax quantisation bf16 -> e4m3 bytes: returns (q uint8 [.., rows, K], scale f32 [.., rows]) with
    x ~= q * scale[..., None];  x [rows, K] or [B, rows, K] (row/batch strided views allowed)."""
    _chk_dev(x, out, scale)
    assert x.dtype == BF16
    xp, ldx, xbs, R, B = _rows_view(x)
    K = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    assert out.dtype == torch.uint8 and scale.dtype == torch.float32 and scale.is_contiguous() and out.shape == x.shape
    op, ldo, obs, _, _ = _rows_view(out)
    L.check(L.lib().tfx_quantize_rows_fp8(xp, ldx, xbs, op, ldo, obs, scale.data_ptr(), R if x.dim() == 3 else 0, R, B, K,
                                          _stream()), "quantize_rows_fp8")
    return out, scale


def gemm_fp8(a: torch.Tensor, a_scale: torch.Tensor, w: torch.Tensor, w_scale: torch.Tensor,
             bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS,
             gelu_from_col: int = 0, gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
             workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = epi((a[b] @ w.T) * a_scale[b][:, None] * w_scale[None, :] + bias) on e4m3 operands (uint8 storage), bf16 out.
    workspace: optional scratch, enables split-K for few-tile shapes (see gemm)."""
    _chk_dev(a, a_scale, w, w_scale, bias, out, gate, res, workspace)
    assert a.dtype == torch.uint8 and w.dtype == torch.uint8 and w.dim() == 2 and w.stride(1) == 1
    assert a_scale.dtype == torch.float32 and w_scale.dtype == torch.float32 and a_scale.is_contiguous() and w_scale.is_contiguous()
    ap, lda, abs_, M, batch = _rows_view(a)
    N, K = w.shape
    assert a.shape[-1] == K and a_scale.numel() == M * batch and w_scale.numel() == N
    if out is None:
        out = torch.empty(*a.shape[:-1], N, dtype=BF16, device=a.device)
    cp, ldc, cbs, M2, b2 = _rows_view(out)
    assert (M2, b2) == (M, batch) and out.shape[-1] == N and out.dtype == BF16
    g = L.GemmArgs()
    g.A, g.lda, g.a_bstride = ap, lda, abs_
    g.W, g.ldw, g.bias = w.data_ptr(), w.stride(0), _p(bias)
    g.C, g.ldc, g.c_bstride = cp, ldc, cbs
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.epilogue, g.gelu_from_col = epilogue, gelu_from_col
    if epilogue in (EPI_BIAS_GATE_RES, EPI_BIAS_RES):
        assert res is not None
        if epilogue == EPI_BIAS_GATE_RES:
            assert gate is not None and gate.stride(-1) == 1
            g.gate = gate.data_ptr()
            g.gate_bstride = gate.stride(0) if gate.dim() == 2 else 0
        rp, ldr, rbs, _, _ = _rows_view(res)
        g.res, g.ldr, g.r_bstride = rp, ldr, rbs
    if workspace is not None:
        g.workspace, g.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    L.check(L.lib().tfx_gemm_fp8(C.byref(g), a_scale.data_ptr(), M if batch > 1 else 0, w_scale.data_ptr(), _stream()), "gemm_fp8")
    return out


def ln_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None,
                eps: float = 1e-6) -> torch.Tensor:
    """LayerNorm(x) * (1 + scale[b]) + shift[b];  x [B,R,D], shift/scale [B,D] (row-strided views allowed)."""
    _chk_dev(x, shift, scale, out)
    assert x.dim() == 3 and x.dtype == BF16
    if out is None:
        out = torch.empty_like(x)
    xp, ldx, xbs, R, B = _rows_view(x)
    op, ldo, obs, _, _ = _rows_view(out)
    assert shift.stride(-1) == 1 and scale.stride(-1) == 1 and shift.stride(0) == scale.stride(0)
    L.check(L.lib().tfx_ln_modulate(xp, ldx, xbs, op, ldo, obs, shift.data_ptr(), scale.data_ptr(), shift.stride(0),
                                    R, B, x.shape[-1], eps, _stream()), "ln_modulate")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LayerNorm with elementwise affine over the last dim of x [.., D] bf16 (contiguous rows): one bf16 rounding."""
    _chk_dev(x, gamma, beta, out)
    assert x.dtype == BF16 and gamma.dtype == BF16 and beta.dtype == BF16 and x.is_contiguous()
    D = x.shape[-1]
    assert gamma.numel() == D and beta.numel() == D
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().tfx_layernorm(x.data_ptr(), D, out.data_ptr(), D, gamma.data_ptr(), beta.data_ptr(), x.numel() // D, D, eps,
                                  _stream()), "layernorm")
    return out


def ln_modulate_fp8(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, eps: float = 1e-6):
    """ln_modulate followed by quantize_rows_fp8 in one pass: returns (q uint8 [B,R,D], scale f32 [B,R])."""
    _chk_dev(x, shift, scale)
    assert x.dim() == 3 and x.dtype == BF16
    xp, ldx, xbs, R, B = _rows_view(x)
    D = x.shape[-1]
    q = torch.empty(B, R, D, dtype=torch.uint8, device=x.device)
    qs = torch.empty(B, R, dtype=torch.float32, device=x.device)
    assert shift.stride(-1) == 1 and scale.stride(-1) == 1 and shift.stride(0) == scale.stride(0)
    L.check(L.lib().tfx_ln_modulate_fp8(xp, ldx, xbs, q.data_ptr(), D, R * D, qs.data_ptr(), R, shift.data_ptr(),
                                        scale.data_ptr(), shift.stride(0), R, B, D, eps, _stream()), "ln_modulate_fp8")
    return q, qs


def rmsnorm_rope_(buf: torch.Tensor, q_off: int, k_off: int, H: int, T: int, wq_img, wk_img, wq_txt, wk_txt,
                  cos: torch.Tensor, sin: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """In place on buf [B,N,ld]: q at columns q_off.., k at k_off.. (H heads of 128)."""
    _chk_dev(buf, wq_img, wk_img, wq_txt, wk_txt, cos, sin)
    assert buf.dim() == 3 and buf.dtype == BF16 and cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
    B, N, _ = buf.shape
    assert cos.shape == (N, 128)
    L.check(L.lib().tfx_rmsnorm_rope(buf.data_ptr(), buf.stride(1), buf.stride(0), q_off, k_off, H, N, T, B,
                                     wq_img.data_ptr(), wk_img.data_ptr(), wq_txt.data_ptr(), wk_txt.data_ptr(),
                                     cos.data_ptr(), sin.data_ptr(), eps, _stream()), "rmsnorm_rope")
    return buf


def gate_residual(x: torch.Tensor, gate: torch.Tensor, res: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """res + bf16(gate[:, None, :] * x): x, res [B,R,D] (row / batch strides allowed), gate [B,D] -> out [B,R,D] (tfx_gate_residual;
    `hidden_states + gate.unsqueeze(1) * attn_output`, transformer_flux.py:733-735, 817-818)."""
    _chk_dev(x, gate, res, out)
    assert x.dim() == 3 and x.dtype == res.dtype == gate.dtype == BF16 and res.shape == x.shape
    B, R, D = x.shape
    assert gate.shape == (B, D) and x.stride(2) == res.stride(2) == gate.stride(1) == 1
    if out is None:
        out = torch.empty(B, R, D, dtype=BF16, device=x.device)
    assert out.shape == x.shape and out.dtype == BF16 and out.stride(2) == 1
    L.check(L.lib().tfx_gate_residual(x.data_ptr(), x.stride(1), x.stride(0), gate.data_ptr(), gate.stride(0), res.data_ptr(),
                                      res.stride(1), res.stride(0), out.data_ptr(), out.stride(1), out.stride(0), R, B, D,
                                      _stream()), "gate_residual")
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: Optional[torch.Tensor] = None,
              scale: Optional[float] = None, score_bound: float = 0.0, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q,k,v: [B,N,H*128] (views with row/batch strides allowed) -> out [B,N,H*128].  score_bound: the caller's promise
    |scale * q . k| <= score_bound (0 = unknown), see tfx_attn_args.  workspace: optional device scratch (>= 69.2 MB) that lets the
    kernel deal (item, key tile) units to the CUs (stream-K, tfx_attn_args.workspace)."""
    _chk_dev(q, k, v, out, workspace)
    B, N, HD = q.shape
    H = HD // 128
    assert HD % 128 == 0 and q.dtype == BF16
    if out is None:
        out = torch.empty(B, N, HD, dtype=BF16, device=q.device)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    a.q_bstride, a.k_bstride, a.v_bstride, a.o_bstride = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.B, a.H, a.N = B, H, N
    a.scale = scale if scale is not None else 128 ** -0.5
    a.score_bound = float(score_bound)
    if workspace is not None:
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    L.check(L.lib().tfx_joint_attention(C.byref(a), _stream()), "joint_attention")
    return out


def euler_step_(v: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, step: int = 0,
                step_ptr: Optional[torch.Tensor] = None, xin: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_dev(v, x, coef, step_ptr, xin)
    if v.dtype != BF16 or x.dtype != BF16 or (xin is not None and xin.dtype != BF16):
        raise TypeError("euler_step_: the HIP scheduler kernels operate on bf16 latents / model outputs")
    assert v.is_contiguous() and x.is_contiguous() and coef.dtype == torch.float32
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    L.check(L.lib().tfx_euler_step(v.data_ptr(), x.data_ptr(), _p(xin), xin.stride(-2) if xin is not None else 0, Cc,
                                   rows, coef.data_ptr(), _p(step_ptr), step, _stream()), "euler_step")
    return x


def amo_step_(v: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, noise: torch.Tensor, step: int = 0,
              step_ptr: Optional[torch.Tensor] = None, xin: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_dev(v, x, coef, noise, step_ptr, xin)
    if v.dtype != BF16 or x.dtype != BF16 or (xin is not None and xin.dtype != BF16):
        raise TypeError("amo_step_: the HIP scheduler kernels operate on bf16 latents / model outputs")
    assert v.is_contiguous() and x.is_contiguous() and noise.is_contiguous() and noise.dtype == torch.float32
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    L.check(L.lib().tfx_amo_step(v.data_ptr(), x.data_ptr(), _p(xin), xin.stride(-2) if xin is not None else 0, Cc,
                                 rows, coef.data_ptr(), _p(step_ptr), step, noise.data_ptr(), _stream()), "amo_step")
    return x


def timestep_embedding(t: torch.Tensor) -> torch.Tensor:
    _chk_dev(t)
    t = t.contiguous().float()
    out = torch.empty(t.numel(), 256, dtype=BF16, device=t.device)
    L.check(L.lib().tfx_timestep_embedding(t.data_ptr(), out.data_ptr(), t.numel(), _stream()), "timestep_embedding")
    return out


def silu(a: torch.Tensor) -> torch.Tensor:
    _chk_dev(a)
    a = a.contiguous()
    out = torch.empty_like(a)
    L.check(L.lib().tfx_silu(a.data_ptr(), out.data_ptr(), a.numel(), _stream()), "silu")
    return out


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _chk_dev(a, b)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    L.check(L.lib().tfx_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "add")
    return out


def scatter_cols_(src: torch.Tensor, dst: torch.Tensor, col0: int) -> torch.Tensor:
    """dst[..., col0:col0+C] = src  (src [.., C] contiguous, dst [.., ld] contiguous)."""
    _chk_dev(src, dst)
    assert src.is_contiguous() and dst.is_contiguous()
    Cc = src.shape[-1]
    L.check(L.lib().tfx_scatter_cols(src.data_ptr(), dst.data_ptr(), src.numel() // Cc, Cc, dst.shape[-1], col0,
                                     _stream()), "scatter_cols")
    return dst


def copy_rows_(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _chk_dev(src, dst)
    sp, sld, sbs, R, B = _rows_view(src)
    dp, dld, dbs, R2, B2 = _rows_view(dst)
    assert (R, B) == (R2, B2) and src.shape[-1] == dst.shape[-1]
    L.check(L.lib().tfx_copy_rows(sp, sld, sbs, dp, dld, dbs, R, src.shape[-1], B, _stream()), "copy_rows")
    return dst


def select_step_(table: torch.Tensor, cur: torch.Tensor, step_ptr: torch.Tensor) -> None:
    L.check(L.lib().tfx_select_step(table.data_ptr(), cur.data_ptr(), cur.numel(), step_ptr.data_ptr(), _stream()),
            "select_step")


def advance_step_(step_ptr: torch.Tensor) -> None:
    L.check(L.lib().tfx_advance_step(step_ptr.data_ptr(), _stream()), "advance_step")


def prof_enable(on: bool) -> None:
    L.check(L.lib().tfx_prof_enable(1 if on else 0), "prof_enable")


def prof_collect(kind: int):
    """(total kernel ms, total algorithmic FLOPs, launches) of the profiled launches of kind 0 (GEMM) / 1 (attention)."""
    ms, fl, n = C.c_double(), C.c_double(), C.c_int()
    L.check(L.lib().tfx_prof_collect(kind, C.byref(ms), C.byref(fl), C.byref(n)), "prof_collect")
    return ms.value, fl.value, n.value


def mfma_peak_probe(operands: torch.Tensor, fp8: bool = False, seconds: float = 2.5) -> dict:
    """tfx_mfma_peak_probe timed over ~`seconds` with HIP events on the current stream: the rate of an MFMA-only kernel on `operands`
    (bf16, or uint8 holding e4m3 codes) at the board's power cap -- TFLOP/s of the LAST launch of a back-to-back series (the first ones run
    inside the power controller's averaging window at the uncapped clock)."""
    _chk_dev(operands)
    nbytes = operands.numel() * operands.element_size()
    st = _stream()
    fl = C.c_double()

    def launch(ktiles):
        L.check(L.lib().tfx_mfma_peak_probe(operands.data_ptr(), nbytes, 1 if fp8 else 0, ktiles, C.byref(fl), st), "mfma_peak_probe")

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); launch(48 * 64); ev[1].record(); torch.cuda.synchronize()
    per_ktile_ms = ev[0].elapsed_time(ev[1]) / (48 * 64)
    kt = max(48, int(250.0 / max(per_ktile_ms, 1e-6)) // 48 * 48)          # ~0.25 s per launch
    n = max(3, int(seconds / 0.25))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        launch(kt)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    tf = [fl.value / (m * 1e-3) / 1e12 for m in ms]
    return {"tflops": tf[-1], "tflops_first_launch": tf[0], "tflops_min": min(tf), "launches": n, "seconds": sum(ms) * 1e-3,
            "dtype": "e4m3" if fp8 else "bf16", "flops_per_launch": fl.value}


ATTENTION_MODES = ("w4_guarded", "w4_valu_rowsum", "w4_lazy_valu", "w4_lazy", "w4_reference_free", "hp", "w16", "other",
                   "streamk_tail")     # the last: launches (already counted under their kernel form) whose last round was dealt as (item, tile) units


def attention_mode_counts(reset: bool = False) -> dict:
    """Attention launches since the last reset by kernel form (tfx_attention_mode_counts): which stream the score bound selected."""
    c = (C.c_int64 * 9)()
    L.lib().tfx_attention_mode_counts(c, 9, 1 if reset else 0)
    return {name: int(c[i]) for i, name in enumerate(ATTENTION_MODES)}


def release_scratch() -> None:
    """Frees what the library allocated behind optional knobs (today: the attention tail-split partials, one buffer per stream that
    ran with attention_tail_split = 1).  Step graphs captured while the knob was on keep dangling pointers: drop them first."""
    L.check(L.lib().tfx_release_scratch(), "release_scratch")


def set_option(name: str, value: int) -> None:
    L.check(L.lib().tfx_set_option(name.encode(), int(value)), "set_option")


_ZERO_PAGE = {}


def zero_page(device) -> torch.Tensor:
    z = _ZERO_PAGE.get(str(device))
    if z is None:
        z = _ZERO_PAGE[str(device)] = torch.zeros(256, dtype=BF16, device=device)
    return z


def conv3x3_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, up: int = 1,
                 pad_lo: int = 1, res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 variant: int = -1) -> torch.Tensor:
    """x [B, inH, inW, Cin] NHWC bf16, w [Cout, 3, 3, Cin] (KRSC) -> [B, H, W, Cout].  up = 2 folds a nearest 2x upsample
    into the gather; stride = 2, pad_lo = 0 is the VAE downsample (pad (0,1,0,1))."""
    _chk_dev(x, w, bias, res, out)
    assert x.is_contiguous() and w.is_contiguous() and x.dtype == BF16
    B, inH, inW, Cin = x.shape
    Cout = w.shape[0]
    if stride == 1:
        H, W = inH * up, inW * up
    else:
        H, W = (inH * up + pad_lo + 1 - 3) // stride + 1, (inW * up + pad_lo + 1 - 3) // stride + 1
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=BF16, device=x.device)
    L.check(L.lib().tfx_conv3x3_nhwc(x.data_ptr(), B, inH, inW, Cin, w.data_ptr(), _p(bias), out.data_ptr(), H, W, Cout,
                                     stride, up, pad_lo, _p(res), zero_page(x.device).data_ptr(), variant, _stream()),
            "conv3x3_nhwc")
    return out


def pair_conv_weights(w: torch.Tensor, bias: Optional[torch.Tensor]):
    """[Cout, 3, 3, Cin] (KRSC), [Cout] -> the pixel-pair form of tfx_conv3x3_pair_nhwc: [2 Cout, 3, 4, Cin], [2 Cout]."""
    Cout, _, _, Cin = w.shape
    wp = torch.zeros(2, Cout, 3, 4, Cin, dtype=w.dtype, device=w.device)
    wp[0, :, :, 0:3] = w
    wp[1, :, :, 1:4] = w
    return wp.view(2 * Cout, 3, 4, Cin).contiguous(), (torch.cat([bias, bias]).contiguous() if bias is not None else None)


def conv3x3_pair_nhwc(x: torch.Tensor, w_pair: torch.Tensor, bias_pair: Optional[torch.Tensor], res: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3 x 3 convolution, stride 1, pad 1, in pixel-pair form (pair_conv_weights): x [B, H, W, Cin], W even -> [B, H, W, Cout]."""
    _chk_dev(x, w_pair, bias_pair, res, out)
    assert x.is_contiguous() and w_pair.is_contiguous() and x.dtype == BF16
    B, H, W, Cin = x.shape
    Cout = w_pair.shape[0] // 2
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=BF16, device=x.device)
    # the kernel addresses res and out as dense [B, H, W, Cout] (no strides in tfx_conv3x3_pair_nhwc)
    assert out.shape == (B, H, W, Cout) and out.dtype == BF16 and out.is_contiguous()
    assert res is None or (res.shape == (B, H, W, Cout) and res.dtype == BF16 and res.is_contiguous())
    L.check(L.lib().tfx_conv3x3_pair_nhwc(x.data_ptr(), B, H, W, Cin, w_pair.data_ptr(), _p(bias_pair), out.data_ptr(), Cout, _p(res),
                                          zero_page(x.device).data_ptr(), _stream()), "conv3x3_pair_nhwc")
    return out


def blend_edge_nhwc_(a: torch.Tensor, b: torch.Tensor, extent: int, axis: int) -> torch.Tensor:
    """Seam blend of two VAE tiles, in place on b (AutoencoderKL.blend_v: axis 1 = rows, blend_h: axis 2 = columns; a, b NHWC bf16
    views [B, H, W, C]): the first `extent` rows / columns of b fade in from the LAST `extent` of a; extent is clamped to both tiles'
    sizes as the reference does, and the weights use the clamped value."""
    _chk_dev(a, b)
    assert a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype == BF16 and a.stride(3) == b.stride(3) == 1 and axis in (1, 2)
    other = 3 - axis
    extent = min(a.shape[axis], b.shape[axis], int(extent))
    assert a.shape[0] == b.shape[0] and a.shape[3] == b.shape[3] and a.shape[other] == b.shape[other], "tiles of one row / column line up"
    if extent <= 0:
        return b
    a0 = a.narrow(axis, a.shape[axis] - extent, extent)
    L.check(L.lib().tfx_blend_edge_nhwc(a0.data_ptr(), a0.stride(0), a0.stride(axis), a0.stride(other), b.data_ptr(), b.stride(0),
                                        b.stride(axis), b.stride(other), b.shape[0], extent, b.shape[other], b.shape[3], _stream()),
            "blend_edge_nhwc")
    return b


def groupnorm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, silu: bool = True,
                   eps: float = 1e-6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [B, ..., C] NHWC bf16 -> GroupNorm(+SiLU)."""
    _chk_dev(x, gamma, beta, out)
    assert x.is_contiguous() and x.dtype == BF16
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    if out is None:
        out = torch.empty_like(x)
    ws = torch.empty(B * ((HW + 1023) // 1024 + 1) * groups * 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().tfx_groupnorm_nhwc(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), B,
                                       HW, C, groups, eps, 1 if silu else 0, _stream()), "groupnorm_nhwc")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# pixel / latent layout steps around the VAE (imageops.hip)
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.uint8: 2}


def _dt(t: torch.Tensor) -> int:
    if t.dtype not in _DT:
        raise TypeError(f"unsupported dtype {t.dtype} (float32, bfloat16 or uint8)")
    return _DT[t.dtype]


def any_negative(x: torch.Tensor, flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """int32 device flag, set when x has a negative element (no host sync)."""
    _chk_dev(x, flag)
    x = x.contiguous()
    if flag is None:
        flag = torch.zeros(1, dtype=torch.int32, device=x.device)
    L.check(L.lib().tfx_any_negative(x.data_ptr(), _dt(x), x.numel(), flag.data_ptr(), _stream()), "any_negative")
    return flag


def prep_image(img: torch.Tensor, mask: Optional[torch.Tensor] = None, norm_mode: int = 0, binarize: bool = True,
               neg_flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Encoder input [B, H, W, 8] bf16 NHWC = bf16(norm(img) * (1 - mask)).  img: [B, C, H, W] float32 / bfloat16, or
    [B, H, W, C] uint8 (value / 255); mask: [Bm, 1, H, W] / [Bm, H, W] float32 or uint8, Bm in {1, B}."""
    _chk_dev(img, mask, neg_flag)
    img = img.contiguous()
    if img.dtype == torch.uint8:
        B, H, W, Cc = img.shape
    else:
        B, Cc, H, W = img.shape
    mb = 1
    if mask is not None:
        mask = mask.contiguous()
        mb = mask.shape[0]
        assert mask.numel() == mb * H * W, "mask must be [Bm, 1, H, W] or [Bm, H, W] at the image size"
    out = torch.empty(B, H, W, 8, dtype=BF16, device=img.device)
    L.check(L.lib().tfx_prep_image(img.data_ptr(), _dt(img), _p(mask), _dt(mask) if mask is not None else 0, out.data_ptr(),
                                   B, Cc, H, W, mb, norm_mode, 1 if binarize else 0, _p(neg_flag), _stream()), "prep_image")
    return out


_RESAMPLE_TABLES = {}


def resample_u8(x: torch.Tensor, size) -> torch.Tensor:
    """PIL.Image.resize((W, H)) (BICUBIC, the default) of uint8 [B, H, W, C] images on the device, bit-identical to Pillow:
    horizontal pass, then vertical pass, each rounding to uint8 (tfx_resample_u8)."""
    from .image_processor import pil_resample_tables
    _chk_dev(x)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.is_contiguous()
    B, H, W, Cc = x.shape
    Ho, Wo = size
    for axis, (n_in, n_out) in ((2, (W, Wo)), (1, (H, Ho))):
        if n_in == n_out:
            continue
        key = (n_in, n_out, str(x.device))
        if key not in _RESAMPLE_TABLES:
            b, k = pil_resample_tables(n_in, n_out)
            _RESAMPLE_TABLES[key] = (torch.from_numpy(b).to(x.device), torch.from_numpy(k).to(x.device))
        b, k = _RESAMPLE_TABLES[key]
        Bc, Hc, Wc, _ = x.shape
        if axis == 2:
            out, outer, inner = torch.empty(Bc, Hc, n_out, Cc, dtype=torch.uint8, device=x.device), Bc * Hc, Cc
        else:
            out, outer, inner = torch.empty(Bc, n_out, Wc, Cc, dtype=torch.uint8, device=x.device), Bc, Wc * Cc
        L.check(L.lib().tfx_resample_u8(x.data_ptr(), out.data_ptr(), b.data_ptr(), k.data_ptr(), k.shape[1], outer, n_in, n_out,
                                        inner, _stream()), "resample_u8")
        x = out
    return x


def rgb_to_grey(x: torch.Tensor) -> torch.Tensor:
    """PIL convert("L") of uint8 [..., 3] -> uint8 [...]."""
    _chk_dev(x)
    assert x.dtype == torch.uint8 and x.shape[-1] == 3 and x.is_contiguous()
    out = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
    L.check(L.lib().tfx_rgb_to_grey_u8(x.data_ptr(), out.data_ptr(), out.numel(), _stream()), "rgb_to_grey")
    return out


def compose_canvas(glyph: torch.Tensor, scene: torch.Tensor, scene_mask_rgb: torch.Tensor, horizontal: bool = False,
                   mask_rgb: bool = False):
    """uint8 [B, gh, gw, 3] glyph images + [B, sh, sw, 3] scenes + the scenes' RGB masks -> (canvas [B, H, W, 3] u8, mask
    [B, H, W] u8): glyph first (top / left), its mask black, PIL's "L" of the RGB mask elsewhere.  mask_rgb: the mask canvas
    keeps its three channels ([B, H, W, 3]) for a resize before the grey conversion."""
    _chk_dev(glyph, scene, scene_mask_rgb)
    for t in (glyph, scene, scene_mask_rgb):
        assert t.dtype == torch.uint8 and t.dim() == 4 and t.shape[-1] == 3 and t.is_contiguous()
    B, gh, gw, _ = glyph.shape
    _, sh, sw, _ = scene.shape
    assert scene.shape[0] == B and scene_mask_rgb.shape == scene.shape
    H, W = (sh, gw + sw) if horizontal else (gh + sh, sw)
    canvas = torch.empty(B, H, W, 3, dtype=torch.uint8, device=glyph.device)
    cmask = torch.empty((B, H, W, 3) if mask_rgb else (B, H, W), dtype=torch.uint8, device=glyph.device)
    L.check(L.lib().tfx_compose_canvas(glyph.data_ptr(), scene.data_ptr(), scene_mask_rgb.data_ptr(), canvas.data_ptr(),
                                       cmask.data_ptr(), B, gh, gw, sh, sw, 1 if horizontal else 0, 1 if mask_rgb else 0,
                                       _stream()), "compose_canvas")
    return canvas, cmask


def pack_mask(mask: torch.Tensor, out: torch.Tensor, col0: int, B: int, H: int, W: int, binarize: bool = True) -> torch.Tensor:
    """out[b, :, col0 : col0 + 256] = packed mask (out: [B, S, ld] bf16)."""
    _chk_dev(mask, out)
    mask = mask.contiguous()
    mb = mask.shape[0]
    assert mask.numel() == mb * H * W and out.dtype == BF16 and out.is_contiguous()
    L.check(L.lib().tfx_pack_mask(mask.data_ptr(), _dt(mask), out.data_ptr(), B, H, W, mb, 1 if binarize else 0,
                                  out.shape[-1], col0, _stream()), "pack_mask")
    return out


def vae_sample_pack(moments: torch.Tensor, eps: Optional[torch.Tensor], out: torch.Tensor, col0: int, shift: float,
                    scale: float) -> torch.Tensor:
    """moments [B, h, w, 2L] NHWC bf16, eps [B, L, h, w] (None: the mode) -> out[b, :, col0 : col0 + 4L]."""
    _chk_dev(moments, eps, out)
    B, h, w, L2 = moments.shape
    assert moments.is_contiguous() and moments.dtype == BF16 and out.dtype == BF16 and out.is_contiguous()
    if eps is not None:
        eps = eps.contiguous()
        assert eps.shape == (B, L2 // 2, h, w)
    L.check(L.lib().tfx_vae_sample_pack(moments.data_ptr(), _p(eps), _dt(eps) if eps is not None else 1, out.data_ptr(), B, h,
                                        w, L2 // 2, shift, scale, out.shape[-1], col0, _stream()), "vae_sample_pack")
    return out


def unpack_latents(lat: torch.Tensor, h: int, w: int, shift: float, scale: float) -> torch.Tensor:
    """lat [B, (h/2)(w/2), 4L] bf16 -> z [B, h, w, L] NHWC bf16 = lat / scale + shift."""
    _chk_dev(lat)
    assert lat.dtype == BF16 and lat.stride(-1) == 1 and lat.dim() == 3
    B, S, C4 = lat.shape
    assert S == (h // 2) * (w // 2) and lat.stride(0) == S * lat.stride(1)
    out = torch.empty(B, h, w, C4 // 4, dtype=BF16, device=lat.device)
    L.check(L.lib().tfx_unpack_latents(lat.data_ptr(), lat.stride(1), out.data_ptr(), B, h, w, C4 // 4, shift, scale,
                                       _stream()), "unpack_latents")
    return out


def postprocess(x: torch.Tensor, C: int, mode: str, denorm: bool = True, crop=None) -> torch.Tensor:
    """x [B, H, W, Cs] NHWC bf16 -> "pt": [B, C, H, W] bf16 | "np": [B, H, W, C] f32 | "u8": [B, H, W, C] uint8 |
    "pt32": [B, C, H, W] f32; crop = (left, top, right, bottom) in pixels keeps only that window (PIL box convention)."""
    _chk_dev(x)
    assert x.is_contiguous() and x.dtype == BF16
    B, H, W, Cs = x.shape
    x0, y0, x1, y1 = crop if crop is not None else (0, 0, W, H)
    Hc, Wc = y1 - y0, x1 - x0
    code = {"pt": 0, "np": 1, "u8": 2, "pt32": 3}[mode]
    if code in (0, 3):
        out = torch.empty(B, C, Hc, Wc, dtype=BF16 if code == 0 else torch.float32, device=x.device)
    else:
        out = torch.empty(B, Hc, Wc, C, dtype=torch.float32 if code == 1 else torch.uint8, device=x.device)
    L.check(L.lib().tfx_postprocess(x.data_ptr(), out.data_ptr(), B, H, W, Cs, C, code, 1 if denorm else 0, y0, x0, Hc, Wc,
                                    _stream()), "postprocess")
    return out


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B, N, C] -> [B, C, N] (bf16)."""
    _chk_dev(x, out)
    assert x.dim() == 3 and x.dtype == BF16 and x.stride(-1) == 1
    B, N, Cc = x.shape
    if out is None:
        out = torch.empty(B, Cc, N, dtype=BF16, device=x.device)
    L.check(L.lib().tfx_transpose(x.data_ptr(), x.stride(1), x.stride(0), out.data_ptr(), out.stride(1), out.stride(0), N, Cc,
                                  B, _stream()), "transpose")
    return out


def row_softmax(s: torch.Tensor, scale: float, out: torch.Tensor) -> torch.Tensor:
    """out[:, :N] = bf16(softmax(scale * s)) over the last dim; s [R, N] fp32 (row-strided), out [R, >= N] bf16."""
    _chk_dev(s, out)
    assert s.dim() == 2 and s.dtype == torch.float32 and s.stride(1) == 1
    assert out.dim() == 2 and out.dtype == BF16 and out.stride(1) == 1 and out.shape[0] == s.shape[0] and out.shape[1] >= s.shape[1]
    L.check(L.lib().tfx_row_softmax(s.data_ptr(), s.stride(0), out.data_ptr(), out.stride(0), s.shape[0], s.shape[1], scale,
                                    _stream()), "row_softmax")
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = a[b] @ w.T as fp32 raw accumulators (no bias / epilogue).  a [M, K] or [B, M, K] bf16, w [N, K] bf16,
    out [.., M, N] fp32 (row-strided views allowed)."""
    _chk_dev(a, w, out)
    assert a.dtype == BF16 and w.dtype == BF16 and w.dim() in (2, 3) and w.stride(-1) == 1
    ap, lda, abs_, M, batch = _rows_view(a)
    N, K = w.shape[-2:]
    assert a.shape[-1] == K and (w.dim() == 2 or (a.dim() == 3 and w.shape[0] == batch)), "w [N, K], or [B, N, K] with a [B, M, K]"
    if out is None:   # rows padded to a multiple of 4 floats (16-byte aligned vector stores)
        out = torch.empty(*a.shape[:-1], (N + 3) // 4 * 4, dtype=torch.float32, device=a.device)[..., :N]
    assert out.dtype == torch.float32
    cp, ldc, cbs, M2, b2 = _rows_view(out)
    assert (M2, b2) == (M, batch) and out.shape[-1] == N
    g = L.GemmArgs()
    g.A, g.lda, g.a_bstride = ap, lda, abs_
    g.W, g.ldw, g.bias = w.data_ptr(), w.stride(-2), None
    g.w_bstride = w.stride(0) if w.dim() == 3 else 0
    g.C, g.ldc, g.c_bstride = cp, ldc, cbs
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.epilogue = EPI_BIAS
    L.check(L.lib().tfx_gemm_bf16_f32(C.byref(g), _stream()), "gemm_f32")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# text-encoder kernels (textenc.hip)
def attention64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, rel_bias: Optional[torch.Tensor] = None,
                causal: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q, k, v: [B, N, H*64] bf16 (row / batch strided views allowed), N <= 512 -> [B, N, H*64].  rel_bias: fp32 [H, 2N-1]
    indexed by key - query + N - 1 (T5); causal: keys after the query masked (CLIP)."""
    _chk_dev(q, k, v, rel_bias, out)
    B, N, HD = q.shape
    H = HD // 64
    assert HD % 64 == 0 and q.dtype == BF16
    if rel_bias is not None:
        assert rel_bias.dtype == torch.float32 and rel_bias.is_contiguous() and rel_bias.shape == (H, 2 * N - 1)
    if out is None:
        out = torch.empty(B, N, HD, dtype=BF16, device=q.device)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
    a.q_bstride, a.k_bstride, a.v_bstride, a.o_bstride = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.B, a.H, a.N, a.scale = B, H, N, scale
    L.check(L.lib().tfx_attention64(C.byref(a), _p(rel_bias), 1 if causal else 0, _stream()), "attention64")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """T5LayerNorm over the last dim of x [rows, D] (fp32 or bf16) -> bf16."""
    _chk_dev(x, w, out)
    assert x.dim() == 2 and x.stride(1) == 1 and w.dtype == BF16 and x.dtype in (torch.float32, BF16)
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    L.check(L.lib().tfx_rmsnorm(x.data_ptr(), 0 if x.dtype == torch.float32 else 1, x.stride(0), w.data_ptr(), out.data_ptr(),
                                out.stride(0), x.shape[0], x.shape[1], eps, _stream()), "rmsnorm")
    return out


def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    _chk_dev(table, ids)
    assert table.dtype == BF16 and table.is_contiguous() and ids.dtype == torch.int64
    ids = ids.contiguous().view(-1)
    out = torch.empty(ids.numel(), table.shape[1], dtype=BF16, device=table.device)
    L.check(L.lib().tfx_gather_rows(table.data_ptr(), ids.data_ptr(), out.data_ptr(), ids.numel(), table.shape[1],
                                    table.shape[0], _stream()), "gather_rows")
    return out


def add_into_f32_(x32: torch.Tensor, y: torch.Tensor, assign: bool = False) -> torch.Tensor:
    _chk_dev(x32, y)
    assert x32.dtype == torch.float32 and x32.is_contiguous() and y.is_contiguous() and y.numel() == x32.numel()
    mode = 2 if assign else (0 if y.dtype == BF16 else 1)
    assert y.dtype in (BF16, torch.float32) and not (assign and y.dtype != BF16)
    L.check(L.lib().tfx_add_into_f32(x32.data_ptr(), y.data_ptr(), x32.numel(), mode, _stream()), "add_into_f32")
    return x32


def mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a * b for bf16 [rows, cols] row-strided views."""
    _chk_dev(a, b)
    assert a.dim() == 2 and a.shape == b.shape and a.stride(1) == 1 and b.stride(1) == 1 and a.dtype == BF16
    out = torch.empty(a.shape, dtype=BF16, device=a.device)
    L.check(L.lib().tfx_mul_act(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), a.shape[0],
                                a.shape[1], 0, _stream()), "mul")
    return out


def quick_gelu(a: torch.Tensor) -> torch.Tensor:
    _chk_dev(a)
    assert a.dim() == 2 and a.stride(1) == 1 and a.dtype == BF16
    out = torch.empty(a.shape, dtype=BF16, device=a.device)
    L.check(L.lib().tfx_mul_act(a.data_ptr(), a.stride(0), None, 0, out.data_ptr(), out.stride(0), a.shape[0], a.shape[1], 1,
                                _stream()), "quick_gelu")
    return out
