"""Build-time check of the emitted gfx950 ISA (no GPU needed: hipcc cross-compiles): no VALU / memory instruction -- the ones
inside inline asm included, which hipcc's own hazard recogniser cannot see -- touches an MFMA result inside the wait-state
window the hardware does not interlock (tools/check_mfma_hazard.py).  The default attention kernel reads its scores from
inline asm; round 2 guarded that by a distance the compiler did not know about.  Now: a compiler-visible read (W4_TOUCH)
in front of the asm reads makes hipcc pad by construction, and this test fails the build if the property is ever lost --
demonstrated on a deliberately shortened distance (-DW4_NO_TOUCH -DW4_HAZARD_SELFTEST)."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "textflux_amd", "csrc")
sys.path.insert(0, os.path.join(REPO, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")


def isa(src, tmp_path, *flags):
    out = tmp_path / (os.path.basename(src) + "".join(f.replace("-", "_") for f in flags) + ".s")
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if src.endswith(("attention_w4.hip", "attention_w16.hip")) else []   # as the Makefile builds them
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S", *extra, *flags,
                        "-o", str(out), os.path.join(CSRC, src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text()


def test_default_attention_kernel_has_no_mfma_result_hazard(tmp_path):
    import check_mfma_hazard as ck
    # product build: attn_w4_kernel<0> (guarded) and <4> (reference-free); bench build (-DTFX_BENCH, round 6): every bookkeeping mode
    for flags, nk in (((), 2), (("-DTFX_BENCH",), 5)):
        asm = isa("attention_w4.hip", tmp_path, *flags)
        ks = [k for k in ck.check_all(asm) if "attn_w4_kernel" in k[0]]      # (the file also holds the tail-split merge kernel)
        assert len(ks) == nk, (flags, [k[0] for k in ks])
        for name, n, n_mfma, rep in ks:
            assert n_mfma > 250 and n > 3000, name                           # the kernel was really parsed
            assert rep == [], (name, rep[:5])
        assert asm.count("v_readfirstlane_b32") >= nk * 16                   # the compiler-visible touches are in the stream


def test_attention_w16_kernel_has_no_mfma_or_transcendental_result_hazard(tmp_path):
    """The 16 x 16 x 32 attention kernel: MFMA results (scores) and v_exp_f32 results are both read from inline asm.  (Its first
    version ordered the exponentials / packs so that hipcc could put a pack right behind the v_exp_f32 it reads: this check found
    it before the GPU did.)"""
    import check_mfma_hazard as ck
    asm = isa("attention_w16.hip", tmp_path, "-DTFX_BENCH")     # bench library only since round 6
    (name, n, n_mfma, rep), = ck.check_all(asm)
    assert "attn_w16_kernel" in name and n_mfma >= 4 * 136
    assert rep == [], rep[:5]


def test_checker_catches_a_shortened_distance(tmp_path):
    """The self-test variant reads a score from inline asm right behind its chain's last MFMA, with the touch removed."""
    import check_mfma_hazard as ck
    asm = isa("attention_w4.hip", tmp_path, "-DTFX_BENCH", "-DW4_NO_TOUCH", "-DW4_HAZARD_SELFTEST")
    reps = [k[3] for k in ck.check_all(asm) if "attn_w4_kernel" in k[0]]
    assert len(reps) == 5
    for rep in reps:
        assert len(rep) >= 4 and all("mfma write" in r for r in rep), rep[:3]
    # ... and the touch alone is what makes the compiler pad: same shortened read, touch in place right behind the MFMA
    padded = isa("attention_w4.hip", tmp_path, "-DTFX_BENCH", "-DW4_HAZARD_SELFTEST")
    reps2 = [k[3] for k in ck.check_all(padded) if "attn_w4_kernel" in k[0]]
    assert len(reps2) == 5 and all(len(r) >= 4 and all("mfma write" in x for x in r) for r in reps2)   # the unprotected asm read is still flagged (the touch sits in front of the REAL reads only)


def test_checker_agrees_with_hipcc_on_visible_instructions(tmp_path):
    """Stripping the s_nop padding hipcc itself inserted in front of compiler-visible VALU reads must trip the checker: the
    rule implemented here is the compiler's rule (passes + 4 wait states on gfx950), not a looser one."""
    import re
    import check_mfma_hazard as ck
    asm = isa("textenc.hip", tmp_path)
    assert all(rep == [] for _, _, _, rep in ck.check_all(asm))
    stripped = re.sub(r"^\s*s_nop \d+\s*$", "", asm, flags=re.M)
    assert any(rep for _, _, _, rep in ck.check_all(stripped))


@pytest.mark.parametrize("src,flags", [("attention.hip", ("-DTFX_BENCH",)), ("attention_hp.hip", ("-DTFX_BENCH",)), ("gemm.hip", ()),
                                       ("gemm.hip", ("-DTFX_BENCH",)), ("textenc.hip", ())])
def test_other_mfma_kernels_are_clean(src, flags, tmp_path):
    """(the 8-wave attention kernels, the half-tile kernel and the 4-wave GEMM are bench-library builds since round 6)"""
    import check_mfma_hazard as ck
    res = ck.check_all(isa(src, tmp_path, *flags))
    assert res and all(rep == [] for _, _, _, rep in res), [(n, r[:2]) for n, _, _, r in res if r]


def test_synthetic_listing_known_answers():
    import check_mfma_hazard as ck
    head = "\t.amdhsa_kernel k\nk:\n"
    ok = head + "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n" + "\ts_nop 11\n\tv_add_f32 v40, v0, v1\n\ts_endpgm\n"
    assert ck.check(ok, "k")[3] == []
    short = ok.replace("s_nop 11", "s_nop 10")
    assert len(ck.check(short, "k")[3]) == 1
    # across a loop back-edge: the MFMA at the bottom of the loop, the read at its top
    loop = head + ".LBB0_1:\n\tv_max_f32 v40, v0, v1\n" + "\ts_nop 3\n" * 2 + "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n" \
        "\ts_cbranch_scc1 .LBB0_1\n\ts_nop 15\n\ts_endpgm\n"
    assert len(ck.check(loop, "k")[3]) == 1
    # a dependent MFMA (SrcC chain) is the compiler's business, an LDS write of the result is not
    chain = head + "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], 0\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n" \
        "\ts_nop 11\n\tds_write_b32 v50, v3\n\ts_endpgm\n"
    assert ck.check(chain, "k")[3] == []
    assert len(ck.check(chain.replace("s_nop 11", "s_nop 2"), "k")[3]) == 1
    # round 6: a DOT result needs three wait states before a VALU reads it (not interlocked on gfx940+); the same opcode may chain it as
    # its accumulator at once.  The q / k-norm epilogue's first version had these instructions in inline asm without the s_nop: a forward
    # was not deterministic, and nothing static noticed
    dot = head + "\tv_dot2_f32_bf16 v14, v10, v46, 0\n\tv_dot2_f32_bf16 v15, v10, v50, 0\n\tv_cvt_pk_bf16_f32 v14, v14, v15\n\ts_endpgm\n"
    assert len(ck.check(dot, "k")[3]) == 2
    assert ck.check(dot.replace("\tv_cvt", "\ts_nop 2\n\tv_cvt"), "k")[3] == []
    chain = head + "\tv_dot2c_f32_bf16_e32 v1, v2, v2\n\tv_dot2c_f32_bf16_e32 v1, v3, v3\n\ts_nop 2\n\tv_add_f32 v4, v1, v1\n\ts_endpgm\n"
    assert ck.check(chain, "k")[3] == []
    assert len(ck.check(chain.replace("s_nop 2", "s_nop 0"), "k")[3]) == 1
    # transcendental result -> VALU: one instruction in between
    tr = head + "\tv_exp_f32_e32 v1, v2\n\tv_cvt_pk_bf16_f32 v3, v1, v4\n\ts_endpgm\n"
    assert len(ck.check(tr, "k")[3]) == 1
    assert ck.check(tr.replace("\tv_cvt", "\ts_nop 0\n\tv_cvt"), "k")[3] == []


# ---------------------------------------------------------------------------------------------------- persistent GEMM
def _kernels(asm, prefix):
    """{mangled name: text} of every kernel whose demangled-ish name contains `prefix` (label .. .Lfunc_end)."""
    import re
    out = {}
    for m in re.finditer(r"^(_ZN3tfx\d+" + prefix + r"\w*):\s.*?^\.Lfunc_end\d+:", asm, re.S | re.M):
        out[m.group(1)] = m.group(0)
    return out


def test_persistent_gemm_keeps_its_k_loop_free_of_spills_and_full_drains(tmp_path):
    """Two properties of the emitted persistent GEMM that cost 2-8 % when they were lost, checked on
    the ISA because nothing functional notices them:
      * no bf16 instantiation spills (scratch = 0): a reload in front of the K loop makes hipcc's wait-count pass protect the
        reloaded register INSIDE the loop with s_waitcnt vmcnt(0), which drains the operand prefetch every iteration;
      * no s_waitcnt vmcnt(0) inside the steady-state K loop of ANY instantiation: the counted waits (vmcnt(10) / (12)) are
        the only VMEM waits there (the epilogue's drain is a builtin the pass can see, not inline asm)."""
    import re
    asm = isa("gemm.hip", tmp_path)
    ks = _kernels(asm, "gemm8pp_kernel")
    assert len(ks) >= 12, sorted(ks)
    for name, text in ks.items():
        fp8 = re.search(r"gemm8pp_kernelILi\dELi\dELb1", name) is not None
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", asm[asm.index(".amdhsa_kernel " + name):]).group(1))
        if not fp8:
            assert scratch == 0, (name, scratch)
        loops = [m.start() for m in re.finditer(r"Inner Loop Header", text)]
        assert loops, name
        for a in loops:
            body = text[a:]
            m = re.search(r"\n\ts_cbranch_\w+ \.LBB\d+_\d+\n", body)       # the loop's back-edge: first branch after the header
            body = body[:m.end()] if m else body
            n_mfma = len(re.findall(r"v_mfma_", body))
            if n_mfma < 64:                                                   # not a K loop
                continue
            assert "s_waitcnt vmcnt(0)" not in body, name
            assert "scratch_" not in body, name
            # round 6: no COMPILER-inserted VMEM wait of any count in a K loop (the hand-placed counted waits are inline asm: ;;#ASMSTART
            # precedes them).  Round 5's fp8 q/k-norm instantiation had three `s_waitcnt vmcnt(4)` there -- hipcc protecting spill reloads
            # of the request offsets -- which vmcnt(0) alone did not catch
            lines = body.split("\n")
            own = [l.strip() for i, l in enumerate(lines) if "s_waitcnt vmcnt" in l and "ASMSTART" not in lines[i - 1]]
            assert own == [], (name, own)
            assert len(re.findall(r"s_waitcnt vmcnt\(1[02]\)", body)) >= 4, name   # the counted waits are there
