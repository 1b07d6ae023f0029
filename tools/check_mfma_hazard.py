#!/usr/bin/env python3
"""Static check of the MFMA-result hazards in emitted gfx950 ISA.

On CDNA3/4 the matrix pipe's results are NOT interlocked against the vector ALU / memory pipes: after an XDL (MFMA) write
of a VGPR tuple a VALU / VMEM / LDS instruction touching any register of that tuple needs passes + 4 wait states on gfx950
(8-pass v_mfma_f32_32x32x16_bf16: 12; this is what LLVM's GCNHazardRecognizer pads with s_nop in front of instructions it
KNOWS to be VALU -- checked against its own output), a transcendental result (v_exp / v_rcp / ...) needs one
instruction before a VALU reads it, and (round 6) a DOT result (v_dot2*_f32_bf16 ...) three wait states before a VALU reads it.  Instructions inside `asm` blocks are opaque to the recogniser, so a kernel that reads
MFMA results from inline asm (attention_w4.hip keeps its row maxima and bf16 packs there so that they stay where the
schedule puts them) is only correct if the DISTANCE happens to be large enough.  This script makes that a build-time
property: it walks the kernel's control-flow graph in the compiler's assembly output and reports every non-MFMA
instruction that touches a register while an MFMA (or transcendental) write to it is still inside its window, on any path
(dataflow to a fixpoint, so loop back-edges and out-of-line branches are covered).

    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o k.s kernel.hip ; python tools/check_mfma_hazard.py k.s [--kernel NAME]

Wait states are counted as LLVM does: every instruction one, `s_nop N` N + 1 (an MFMA that waits for the busy pipe takes
longer in reality; the check is the conservative, documented rule).  MFMA -> MFMA dependences (SrcC chains, operand reads)
are the compiler's own business (the intrinsics are visible to it) and are not checked here.  Exit status 1 on a violation.
"""
from __future__ import annotations

import argparse
import re
import sys
from typing import Dict, List, Optional, Set, Tuple

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
REG_RE = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def mfma_passes(mn: str) -> int:
    """Passes (4 cycles each) of the MFMA `mn` on gfx950; unknown shapes are taken as 16 (the longest)."""
    m = re.search(r"_(\d+)x(\d+)x(\d+)", mn)
    if not m:
        return 16
    a, b, k = (int(x) for x in m.groups())
    if "f8f6f4" in mn or "_f8" in mn or "_bf8" in mn or "_fp8" in mn:
        return 16 if a == 32 else 8          # 32x32x64: 16 passes; 16x16x128: 8
    if "f64" in mn:
        return 16
    if a == 32 and b == 32:
        return {16: 8, 8: 16, 4: 16, 2: 16, 1: 16}.get(k, 16)     # 32x32x16 bf16/f16: 8 passes on gfx950; 32x32x8: 16
    if a == 16 and b == 16:
        return {32: 4, 16: 8, 8: 8, 4: 8, 1: 8}.get(k, 8)         # 16x16x32: 4; 16x16x16: 8
    if a == 4:
        return 2
    return 16


def regs_of(opnd: str) -> Set[Tuple[str, int]]:
    out: Set[Tuple[str, int]] = set()
    for m in REG_RE.finditer(opnd):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("line", "text", "mn", "ops", "label_target", "kind", "wait", "dst", "touch")

    def __init__(self, line: int, text: str):
        self.line, self.text = line, text
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.label_target: Optional[str] = None
        self.wait = 1
        if self.mn == "s_nop":
            self.wait = int(self.ops[0], 0) + 1
        if self.mn.startswith("s_cbranch") or self.mn == "s_branch":
            self.label_target = self.ops[-1]
        is_mfma = self.mn.startswith(("v_mfma", "v_smfmac"))
        self.kind = "mfma" if is_mfma else "trans" if self.mn.startswith(TRANS) else "dot" if self.mn.startswith("v_dot") else "other"
        self.dst = regs_of(self.ops[0]) if (self.ops and self.kind in ("mfma", "trans", "dot")) else set()
        # every vector / memory instruction that names a register may read or overwrite it; scalar instructions cannot
        touches = self.mn.startswith(("v_", "ds_", "buffer_", "global_", "flat_", "scratch_", "exp", "image_", "tbuffer_"))
        self.touch = set().union(*(regs_of(o) for o in self.ops)) if (touches and self.ops) else set()


def parse_kernel(asm: str, kernel: Optional[str]) -> Tuple[str, List[Ins], Dict[str, int]]:
    """(name, instructions, label -> index of the next instruction) of the kernel whose symbol contains `kernel` (or the
    first .amdhsa kernel body found)."""
    lines = asm.splitlines()
    syms = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, flags=re.M))
    start = name = None
    for i, ln in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not m.group(1).startswith((".L", "__")) and (kernel is None or kernel == m.group(1) or
                                                              (kernel not in syms and kernel in m.group(1))):
            # a function label: followed (eventually) by instructions up to s_endpgm
            start, name = i + 1, m.group(1)
            break
    if start is None:
        raise SystemExit(f"kernel {kernel!r} not found")
    ins: List[Ins] = []
    labels: Dict[str, int] = {}
    for i in range(start, len(lines)):
        ln = lines[i].split(";", 1)[0].strip() if not lines[i].lstrip().startswith(";") else ""
        if not ln:
            continue
        m = re.match(r"^([.\w$]+):$", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if ln.startswith("."):
            if ln.startswith((".Lfunc_end", ".section", ".size")) and ins:
                break
            continue
        ins.append(Ins(i + 1, ln))
    return name, ins, labels


def check(asm: str, kernel: Optional[str] = None, margin: int = 0, verbose: bool = False):
    name, ins, labels = parse_kernel(asm, kernel)
    n = len(ins)
    succ: List[List[int]] = []
    for i, x in enumerate(ins):
        s: List[int] = []
        if x.mn == "s_endpgm":
            pass
        elif x.mn == "s_branch":
            s.append(labels[x.label_target])
        else:
            if i + 1 < n:
                s.append(i + 1)
            if x.label_target is not None and x.label_target in labels:
                s.append(labels[x.label_target])
        succ.append([t for t in s if t < n])
    # state BEFORE instruction i: {reg: (remaining wait states, line of the producer, kind)}
    state: List[Dict[Tuple[str, int], Tuple[int, int, str]]] = [dict() for _ in range(n)]
    seen = [False] * n
    work = [0]
    seen[0] = True
    violations = {}
    ins_by_line = {x.line: x for x in ins}
    while work:
        i = work.pop()
        x = ins[i]
        st = state[i]
        if x.kind != "mfma":
            for r in x.touch:
                if r in st:
                    rem, src, kind = st[r]
                    if kind == "trans" and (x.kind == "trans" or not x.mn.startswith("v_")):
                        continue      # TRANS -> TRANS forwards; memory instructions read the register file later
                    if kind == "dot":
                        # round 6.  A DOT (v_dot2*_f32_bf16 ...) result is not interlocked on gfx940+: 3 wait states before a VALU reads it
                        # (LLVM GCNHazardRecognizer: DotWriteDifferentVALURead / DotWriteSameDotReadSrcAB = 3).  Free: the same opcode
                        # taking it as its accumulator (SrcC: the destination of the VOP2 `c` forms, the fourth operand of the VOP3P forms)
                        if not x.mn.startswith("v_"):
                            continue
                        if x.kind == "dot" and x.mn == ins_by_line[src].mn:
                            acc_ops = x.ops[0:1] if "dot2c" in x.mn or x.mn.rstrip("_e32").endswith("c") else x.ops[3:4]
                            other = set().union(*(regs_of(o) for k, o in enumerate(x.ops) if o not in acc_ops and k != 0)) if x.ops else set()
                            if r not in other:
                                continue
                    violations[(x.line, src)] = (x, r, rem, kind)
        out = {}
        for r, (rem, src, kind) in st.items():
            if rem - x.wait > 0:
                out[r] = (rem - x.wait, src, kind)
        if x.kind == "mfma":
            req = mfma_passes(x.mn) + 4 + margin
            for r in x.dst:
                out[r] = (req, x.line, "mfma")
        elif x.kind == "trans":
            for r in x.dst:
                out[r] = (1, x.line, "trans")
        elif x.kind == "dot":
            for r in x.dst:
                out[r] = (3, x.line, "dot")
        else:     # an ordinary write retires older pending entries of the same register only after it passed the check above
            pass
        for t in succ[i]:
            tgt = state[t]
            changed = not seen[t]
            for r, v in out.items():
                if r not in tgt or tgt[r][0] < v[0]:
                    tgt[r] = v
                    changed = True
            if changed:
                seen[t] = True
                work.append(t)
    rep = []
    for (line, src), (x, r, rem, kind) in sorted(violations.items()):
        rep.append(f"line {line}: `{x.text}` touches {r[0]}{r[1]} {rem} wait state(s) too early after the {kind} write at line {src}")
    n_mfma = sum(1 for x in ins if x.kind == "mfma")
    if verbose:
        print(f"{name}: {n} instructions, {n_mfma} MFMAs, {len(rep)} hazard violation(s)")
    return name, n, n_mfma, rep


def kernel_symbols(asm: str) -> List[str]:
    return re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, flags=re.M)


def check_all(asm: str, margin: int = 0, verbose: bool = False):
    """Every kernel of the file: [(symbol, instructions, MFMAs, violations)]."""
    return [check(asm, k, margin, verbose) for k in kernel_symbols(asm)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default=None, help="substring of the kernel symbol (default: every kernel of the file)")
    ap.add_argument("--margin", type=int, default=0, help="extra wait states demanded on top of passes + 4")
    a = ap.parse_args()
    with open(a.asm) as f:
        asm = f.read()
    res = [check(asm, a.kernel, a.margin, verbose=True)] if a.kernel else check_all(asm, a.margin, verbose=True)
    bad = 0
    for name, _, _, rep in res:
        for r in rep:
            print("HAZARD", name, r)
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
