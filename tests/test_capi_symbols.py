"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
include/textflux_hip.h declares (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

from textflux_amd import _lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L.build()
    return L.lib()


def header_symbols():
    src = open(os.path.join(REPO, "include", "textflux_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_bound_and_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 17
    assert sorted(L.SIGNATURES) == syms
    for s in syms:
        assert hasattr(lib, s)


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.tfx_version()
    assert isinstance(lib.tfx_last_error(), bytes)


def test_abi_stamp_matches_the_binding_and_a_foreign_library_is_refused(lib, monkeypatch):
    """a libtextflux_hip.so that arrives without its source hash is not rebuilt -- so it must prove it was built
    from THIS header: tfx_abi_info = {TFX_ABI_VERSION, sizeof the four argument structs}, compared in _lib.lib()."""
    import ctypes as C
    got = (C.c_int32 * 7)()
    assert lib.tfx_abi_info(got, 7) == 7
    assert L.ABI_VERSION == L.header_abi_version()      # the binding's constant and the header move together
    assert list(got) == [L.ABI_VERSION, C.sizeof(L.GemmArgs), C.sizeof(L.AttnArgs), C.sizeof(L.DitDesc), C.sizeof(L.StepDesc),
                         C.sizeof(L.DoubleBlock), C.sizeof(L.SingleBlock)]
    two = (C.c_int32 * 2)(-1, -1)
    assert lib.tfx_abi_info(two, 1) == 7 and list(two) == [L.ABI_VERSION, -1]     # writes only what fits
    L._check_abi(lib)
    monkeypatch.setattr(L, "ABI_VERSION", L.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI stamp"):
        L._check_abi(lib)                               # a library of another header version
    monkeypatch.undo()
    # the load-time check needs no header file (a binding copied next to a hand-built .so): only the test above reads it
    monkeypatch.setattr(L, "header_abi_version", lambda: (_ for _ in ()).throw(RuntimeError("no header here")))
    L._check_abi(lib)
    monkeypatch.undo()

    class Grown(C.Structure):
        _fields_ = list(L.StepDesc._fields_) + [("new_tail_field", C.c_void_p)]
    monkeypatch.setattr(L, "StepDesc", Grown)
    with pytest.raises(RuntimeError, match="ABI stamp"):
        L._check_abi(lib)                               # the binding grew a struct the library does not know

    class NoStamp:
        def __getattr__(self, name):
            raise AttributeError(name)
    with pytest.raises(RuntimeError, match="predates the ABI stamp"):
        L._check_abi(NoStamp())


def test_null_arguments_are_rejected_without_a_gpu(lib):
    assert lib.tfx_gemm_bf16(None, -1, None) != 0
    assert b"null" in lib.tfx_last_error()
    assert lib.tfx_dit_forward(None, None) != 0
    assert lib.tfx_joint_attention(None, None) != 0
    assert lib.tfx_gemm_bf16_qkn(None, None, None) != 0 and b"null" in lib.tfx_last_error()        # ABI 7 entry points
    assert lib.tfx_mfma_peak_probe(None, 0, 0, 48, None, None) != 0 and b"mfma_peak_probe" in lib.tfx_last_error()


def test_workspace_layout_is_host_arithmetic(lib):
    import ctypes as C
    off, gws = (C.c_int64 * 6)(), C.c_int64()
    B, S, T, D = 8, 4096, 512, 3072
    assert lib.tfx_workspace_layout(B, S, T, D, 0, off, C.byref(gws)) == 0
    N = S + T
    assert list(off)[:3] == [0, B * N * D * 2, 2 * B * N * D * 2] and off[3] == off[4] == -1 and gws.value == 128 << 20      # split-K partials + (round 6) the attention stream-K partials
    assert off[5] == 2 * B * N * D * 2 + B * N * 7 * D * 2
    assert lib.tfx_workspace_bytes(B, S, T, D, 0) == off[5] + gws.value
    assert lib.tfx_workspace_layout(B, S, T, D, 4, off, C.byref(gws)) == 0 and off[3] > 0 and off[4] == off[3] + B * N * 5 * D
    assert all(o % 256 == 0 for o in off) and lib.tfx_workspace_bytes(0, S, T, D, 0) == -1
    assert lib.tfx_dit_step_run(None, None) != 0 and lib.tfx_dit_step_replay(None, None) != 0 and lib.tfx_graph_destroy(None) == 0


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof as gcc sees include/textflux_hip.h == the ctypes mirror in textflux_amd/_lib.py."""
    import ctypes as C
    import subprocess
    structs = {"tfx_gemm_args": L.GemmArgs, "tfx_attn_args": L.AttnArgs, "tfx_linear": L.Linear,
               "tfx_double_block": L.DoubleBlock, "tfx_single_block": L.SingleBlock, "tfx_dit_desc": L.DitDesc,
               "tfx_step_desc": L.StepDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "textflux_hip.h"', 'int main(void){']
    for cname, ct in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "textflux_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_no_torch_compute_left_in_the_package():
    """the convolution / normalisation / attention arithmetic of the path runs on the library's kernels, not
    on torch's (round 1 still routed the VAE's conv_in, mid-block attention and tiny configs through torch / MIOpen)."""
    import glob
    import re
    banned = re.compile(r"F\.conv2d|F\.group_norm|F\.layer_norm|scaled_dot_product_attention|torch\.nn\.functional\.(conv2d|group_norm|layer_norm)")
    for path in glob.glob(os.path.join(REPO, "textflux_amd", "*.py")):
        with open(path) as f:
            for i, line in enumerate(f, 1):
                code = line.split("#")[0]
                assert not banned.search(code), f"{path}:{i}: {line.strip()}"
