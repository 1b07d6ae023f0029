"""ctypes binding of libtextflux_hip.so (include/textflux_hip.h).

The shared library is the product: there is NO fallback.  If it is missing or fails to load, importing the
ops raises -- a CPU or eager-PyTorch path would silently void every parity claim.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# TFX_LIB (tools/ only): path of the -DTFX_BENCH build with every A/B kernel variant (`make -C textflux_amd/csrc bench`)
LIB_PATH = os.environ.get("TFX_LIB") or os.path.join(_HERE, "libtextflux_hip.so")
HASH_PATH = LIB_PATH + ".srchash"
CSRC = os.path.join(_HERE, "csrc")

c_void_p, c_int, c_int32, c_int64, c_float, c_char_p = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_char_p

# TFX_ABI_VERSION of the include/textflux_hip.h the ctypes mirrors below were written against (tests/test_capi_symbols.py asserts
# that it equals the header's): the library's stamp is compared with THIS constant, so a binding copied without include/ still
# loads, and a ctypes mirror edited without the header (or the other way round) fails a test instead of passing the check.
ABI_VERSION = 8


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("a_bstride", c_int64),
        ("W", c_void_p), ("ldw", c_int64),
        ("bias", c_void_p),
        ("C", c_void_p), ("ldc", c_int64), ("c_bstride", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32), ("batch", c_int32),
        ("epilogue", c_int32), ("gelu_from_col", c_int32),
        ("gate", c_void_p), ("gate_bstride", c_int64),
        ("res", c_void_p), ("ldr", c_int64), ("r_bstride", c_int64),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("w_bstride", c_int64),
    ]


class QknArgs(C.Structure):
    _fields_ = [("norm_q", c_void_p), ("norm_k", c_void_p), ("rope_cs", c_void_p), ("pos0", c_int32), ("q0", c_int32), ("q1", c_int32),
                ("k0", c_int32), ("k1", c_int32), ("eps", c_float)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p),
        ("ldq", c_int64), ("ldk", c_int64), ("ldv", c_int64), ("ldo", c_int64),
        ("q_bstride", c_int64), ("k_bstride", c_int64), ("v_bstride", c_int64), ("o_bstride", c_int64),
        ("B", c_int32), ("H", c_int32), ("N", c_int32), ("scale", c_float), ("score_bound", c_float),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
    ]


class Linear(C.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("w8", c_void_p), ("w8_scale", c_void_p)]


class DoubleBlock(C.Structure):
    _fields_ = [(n, Linear) for n in ("qkv_img", "qkv_txt", "out_img", "out_txt", "ff1_img", "ff2_img", "ff1_txt",
                                      "ff2_txt")] + [(n, c_void_p) for n in ("norm_q", "norm_k", "norm_added_q",
                                                                             "norm_added_k")] + [("attn_score_bound", c_float)]


class SingleBlock(C.Structure):
    _fields_ = [("qkv_mlp", Linear), ("proj_out", Linear), ("norm_q", c_void_p), ("norm_k", c_void_p), ("attn_score_bound", c_float)]


class DitDesc(C.Structure):
    _fields_ = [
        ("D", c_int32), ("H", c_int32), ("in_channels", c_int32), ("out_channels", c_int32),
        ("n_double", c_int32), ("n_single", c_int32),
        ("B", c_int32), ("S", c_int32), ("T", c_int32),
        ("x_embedder", Linear), ("proj_out", Linear),
        ("dbl", C.POINTER(DoubleBlock)), ("sgl", C.POINTER(SingleBlock)),
        ("xin", c_void_p), ("ctx0", c_void_p),
        ("mod", c_void_p), ("mod_bstride", c_int64),
        ("cos_tab", c_void_p), ("sin_tab", c_void_p),
        ("hid", c_void_p), ("xn", c_void_p), ("y", c_void_p),
        ("out", c_void_p),
        ("first_block", c_int32), ("last_block", c_int32), ("flags", c_int32),
        ("q8", c_void_p), ("q8_scale", c_void_p),
        ("gemm_workspace", c_void_p), ("gemm_workspace_bytes", c_int64),
        ("rope_cs", c_void_p),
        ("euler_gate", c_void_p), ("euler_gate_bstride", c_int64),
        ("attn_score_bound", c_float),
    ]


class StepDesc(C.Structure):
    _fields_ = [("dit", DitDesc), ("mod_table", c_void_p), ("mod_cur", c_void_p), ("mod_step_elems", c_int64),
                ("step_ptr", c_void_p), ("latents", c_void_p), ("coef", c_void_p), ("noise", c_void_p), ("sampler", c_int32)]


# name -> (restype, argtypes); every symbol include/textflux_hip.h declares must appear here (tests check it)
SIGNATURES = {
    "tfx_version": (c_char_p, []),
    "tfx_last_error": (c_char_p, []),
    "tfx_abi_info": (c_int, [C.POINTER(c_int32), c_int]),
    "tfx_query_arch": (c_int, [c_char_p, c_int]),
    "tfx_gemm_bf16": (c_int, [C.POINTER(GemmArgs), c_int, c_void_p]),
    "tfx_gemm_bf16_qkn": (c_int, [C.POINTER(GemmArgs), C.POINTER(QknArgs), c_void_p]),
    "tfx_gemm_bf16_f32": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "tfx_gemm_fp8": (c_int, [C.POINTER(GemmArgs), c_void_p, c_int64, c_void_p, c_void_p]),
    "tfx_ln_modulate_fp8": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_int64, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "tfx_quantize_rows_fp8": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                      c_int32, c_void_p]),
    "tfx_ln_modulate": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                c_int32, c_int32, c_int32, c_float, c_void_p]),
    "tfx_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "tfx_rmsnorm_rope": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "tfx_rmsnorm_rope_qk": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "tfx_gate_residual": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                  c_int32, c_int32, c_int32, c_void_p]),
    "tfx_blend_edge_nhwc": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "tfx_joint_attention": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "tfx_euler_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_int32,
                               c_void_p]),
    "tfx_amo_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_int32,
                             c_void_p, c_void_p]),
    "tfx_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "tfx_silu": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "tfx_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "tfx_scatter_cols": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int32, c_void_p]),
    "tfx_copy_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
                              c_void_p]),
    "tfx_select_step": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "tfx_advance_step": (c_int, [c_void_p, c_void_p]),
    "tfx_dit_forward": (c_int, [C.POINTER(DitDesc), c_void_p]),
    "tfx_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "tfx_workspace_layout": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, C.POINTER(c_int64), C.POINTER(c_int64)]),
    "tfx_dit_step_run": (c_int, [C.POINTER(StepDesc), c_void_p]),
    "tfx_dit_step_capture": (c_int, [C.POINTER(StepDesc), c_void_p, C.POINTER(c_void_p)]),
    "tfx_dit_step_replay": (c_int, [c_void_p, c_void_p]),
    "tfx_graph_destroy": (c_int, [c_void_p]),
    "tfx_conv3x3_nhwc": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                 c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int, c_void_p]),
    "tfx_conv3x3_pair_nhwc": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "tfx_groupnorm_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
                                   c_float, c_int32, c_void_p]),
    "tfx_any_negative": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "tfx_prep_image": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                               c_int32, c_int32, c_void_p, c_void_p]),
    "tfx_compose_canvas": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_int32, c_void_p]),
    "tfx_rgb_to_grey_u8": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "tfx_resample_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "tfx_pack_mask": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                              c_void_p]),
    "tfx_vae_sample_pack": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                    c_float, c_int64, c_int32, c_void_p]),
    "tfx_unpack_latents": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float,
                                   c_void_p]),
    "tfx_postprocess": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                c_int32, c_int32, c_void_p]),
    "tfx_transpose": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "tfx_row_softmax": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "tfx_attention64": (c_int, [C.POINTER(AttnArgs), c_void_p, c_int32, c_void_p]),
    "tfx_rmsnorm": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_float, c_void_p]),
    "tfx_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p]),
    "tfx_add_into_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "tfx_mul_act": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p]),
    "tfx_set_option": (c_int, [c_char_p, c_int]),
    "tfx_release_scratch": (c_int, []),
    "tfx_prof_enable": (c_int, [c_int]),
    "tfx_prof_collect": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int)]),
    "tfx_mfma_peak_probe": (c_int, [c_void_p, c_int64, c_int32, c_int32, C.POINTER(C.c_double), c_void_p]),
    "tfx_attention_mode_counts": (c_int, [C.POINTER(c_int64), c_int32, c_int32]),
    "tfx_debug_attention_timing": (c_int, [c_void_p]),
}

_lib = None


def _sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", "Makefile")))
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "textflux_hip.h"))
    return srcs


def _src_hash() -> str:
    h = hashlib.sha256()
    for p in _sources():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def header_abi_version() -> int:
    """TFX_ABI_VERSION of include/textflux_hip.h (tests compare it with ABI_VERSION; lib() does not need the header)."""
    import re
    path = os.path.join(os.path.dirname(_HERE), "include", "textflux_hip.h")
    try:
        with open(path) as f:
            m = re.search(r"#define\s+TFX_ABI_VERSION\s+(\d+)", f.read())
    except OSError as e:
        raise RuntimeError(f"cannot read {path} ({e}): the header ships with the package sources") from e
    if m is None:
        raise RuntimeError(f"{path} does not define TFX_ABI_VERSION")
    return int(m.group(1))


def _check_abi(l: C.CDLL) -> None:
    """A library that does not export tfx_abi_info, or reports another TFX_ABI_VERSION / other struct sizes than this binding's
    ctypes mirrors, was built from a different header: refuse it (a grown struct's tail fields would be ignored silently)."""
    try:
        fn = l.tfx_abi_info
    except AttributeError:
        raise RuntimeError(f"{LIB_PATH} predates the ABI stamp (no tfx_abi_info): rebuild it (`make -C textflux_amd/csrc -B`)") from None
    fn.restype, fn.argtypes = c_int, [C.POINTER(c_int32), c_int]
    got = (c_int32 * 7)()
    n = fn(got, 7)
    want = [ABI_VERSION, C.sizeof(GemmArgs), C.sizeof(AttnArgs), C.sizeof(DitDesc), C.sizeof(StepDesc), C.sizeof(DoubleBlock),
            C.sizeof(SingleBlock)]
    if n != 7 or list(got) != want:
        raise RuntimeError(f"{LIB_PATH} was built from another include/textflux_hip.h: ABI stamp {list(got)[:n]} != the binding's "
                           f"{want} (version, sizeof gemm_args / attn_args / dit_desc / step_desc / double_block / single_block); rebuild it")


def is_stale() -> bool:
    """True when libtextflux_hip.so is missing or was not built from the sources now in csrc/ (content hash kept in a
    side file next to the library; mtimes do not survive the copy to the GPU box).  A library that arrives WITHOUT its
    hash file (the file is git-ignored; a hand-copied build) is not recompiled behind the caller's back and stays usable where
    hipcc is absent -- but it is not trusted blindly either: lib() checks its ABI stamp (tfx_abi_info: TFX_ABI_VERSION and the
    struct sizes) against this binding and refuses a library built from another header."""
    if os.environ.get("TFX_LIB"):
        return not os.path.exists(LIB_PATH)          # an explicitly chosen build is the caller's business
    if not os.path.exists(LIB_PATH):
        return True
    if not os.path.exists(HASH_PATH):
        return False
    with open(HASH_PATH) as f:
        return f.read().strip() != _src_hash()


def build(force: bool = False) -> str:
    """Compile textflux_amd/csrc for gfx950 into libtextflux_hip.so (hipcc cross-compiles without a GPU).  Serialised by a
    file lock (N torchrun ranks importing at once build once); a hash-triggered rebuild is unconditional (`make -B`): make
    itself compares mtimes, which say nothing after a copy, and would leave a stale library next to a fresh hash."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            stale = is_stale()                        # re-checked under the lock: another rank may have just built it
            if force or stale:
                r = subprocess.run(["make", "-C", CSRC, "-j8", "-B"], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("building libtextflux_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
                with open(HASH_PATH, "w") as f:
                    f.write(_src_hash() + "\n")
            elif not os.path.exists(HASH_PATH) and not os.environ.get("TFX_LIB"):
                pass                                  # shipped library without a hash: left alone (see is_stale)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises if it is not there -- there is no non-HIP product path."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64; it must be in the process BEFORE our library is dlopen'ed so both share
        # ONE HIP runtime (loading ours first pulls /opt/rocm's copy and its kernels then see "no device").
        import torch  # noqa: F401
        if is_stale():
            try:    # same HIP sources, not compiled yet or edited since (never a silently stale .so): compile, never substitute
                build()
            except Exception as e:
                raise RuntimeError(f"{LIB_PATH} is missing or stale and could not be built ({e}); textflux_amd has no "
                                   "CPU / eager fallback") from e
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  textflux_amd has no CPU / eager fallback.")
        l = C.CDLL(LIB_PATH)
        _check_abi(l)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"textflux_hip {what} failed: {lib().tfx_last_error().decode()}")
