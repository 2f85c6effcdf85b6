#!/usr/bin/env python3
"""Static audit of MFMA (and dot-product) result hazards in hipcc's generated assembly (tests/test_host_cpu.py runs it on
dg_corr2.hip; the dot rule on dg_small.hip, the only other file with an arithmetic instruction in inline asm).

Why: every MFMA of k_corr2 is `asm volatile` with literal register operands, so hipcc's hazard recogniser does not know that the
destination registers of such an instruction are not readable for a number of wait states.  Round 4 found a compiler-generated
`v_mov` one `s_nop 0` behind an asm MFMA that silently dropped that MFMA's contribution (profiles/r04_c5bias_*.txt); this
walks the generated code and reports every instance of that class.

Model (gfx950 / CDNA4 ISA "MFMA hazards"; wait states are counted the way the hardware's issue logic sees them):
  * an instruction issues one wait state after the previous one; `s_nop N` is N + 1 wait states;
  * the matrix pipe is in order: an MFMA of `passes` passes keeps it busy for `passes` wait states, so an MFMA that FOLLOWS an
    MFMA issues no earlier than `passes` wait states after it (this is what makes "read the accumulators two MFMAs later" legal,
    the kernel's normal pattern); v_mfma_*_32x32x16_{bf16,f16} is 8 passes on gfx950 (32 clocks), 16x16x32 is 4;
  * a non-MFMA instruction (VALU, v_accvgpr_read, a store / LDS write / export that reads the register as data, or anything that
    WRITES it) may touch a destination register of an MFMA only `REQUIRED` wait states after that MFMA issued; an MFMA that
    reads it as SrcA / SrcB likewise.  SrcC == the previous MFMA's vDst (a dependent accumulate) needs none.
    REQUIRED = 18: the kernel's own rule, which covers the architectural 8 + 2 (+ 1 on gfx950) with margin.
  * control flow: from each MFMA every path is followed (both arms of a conditional branch, the target of s_branch) until
    REQUIRED wait states have passed.
"""
import re
import sys

REQUIRED = 18
PASSES = {"32x32x16": 8, "16x16x32": 4, "32x32x8": 8, "16x16x16": 4, "32x32x2": 16, "32x32x1": 16, "16x16x4": 8, "16x16x1": 8,
          "4x4x4": 2, "32x32x4": 16, "16x16x8": 8, "32x32x64": 16, "16x16x128": 8}

_reg = re.compile(r"(?<![\w.])([va])(?:\[(\d+):(\d+)\]|(\d+))(?![\w\[])")
_label = re.compile(r"^([.\w$]+):")
_branch = re.compile(r"^s_(c?branch)\w*\s+([.\w$]+)")


def regs_of(operand_text):
    """set of ('v' | 'a', index) named in an operand string"""
    out = set()
    for m in _reg.finditer(operand_text):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(text):
    """top-level comma split (brackets kept together)"""
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


class Ins:
    __slots__ = ("idx", "line_no", "text", "mnem", "ops", "is_mfma", "passes", "dst", "ab", "c", "touch", "wait", "target", "kind")


def parse(asm_text, kernel=None):
    """-> (list of Ins, {label: index}).  `kernel`: only the body of that function (substring match of its symbol)."""
    ins, labels = [], {}
    active = kernel is None
    for no, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";") else ""
        if not line:
            continue
        m = _label.match(line)
        if m:
            name = m.group(1)
            if kernel is not None and not name.startswith(".L"):
                active = kernel in name
            if active:
                labels[name] = len(ins)
            continue
        if not active or line.startswith("."):
            if line.startswith(".end_amdhsa_kernel") or line.startswith(".section"):
                pass
            continue
        sp = line.split(None, 1)
        mn = sp[0]
        if not re.match(r"^[a-z_][\w.]*$", mn):
            continue
        I = Ins()
        I.idx, I.line_no, I.text, I.mnem = len(ins), no, line, mn
        I.ops = split_operands(sp[1]) if len(sp) > 1 else []
        I.is_mfma = mn.startswith("v_mfma") or mn.startswith("v_smfmac")
        I.passes, I.dst, I.ab, I.c = 0, set(), set(), set()
        I.target, I.kind = None, None
        I.wait = 1
        if mn == "s_nop" and I.ops:
            I.wait = int(I.ops[0], 0) + 1
        b = _branch.match(line)
        if b:
            I.kind, I.target = b.group(1), b.group(2)
        if mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            I.kind = "end"
        if I.is_mfma:
            shape = re.search(r"_(\d+x\d+x\d+)", mn)
            I.passes = PASSES.get(shape.group(1), 16) if shape else 16
            I.dst = regs_of(I.ops[0])
            I.ab = regs_of(I.ops[1]) | regs_of(I.ops[2])
            I.c = regs_of(I.ops[3]) if len(I.ops) > 3 else set()
            I.touch = set()
        else:
            I.touch = set().union(*[regs_of(o) for o in I.ops]) if I.ops else set()
        ins.append(I)
    return ins, labels


def audit(asm_text, kernel=None, required=REQUIRED):
    """-> (violations, stats).  A violation is (mfma Ins, offending Ins, wait states between their issues)."""
    ins, labels = parse(asm_text, kernel)
    bad, n_mfma, min_seen = [], 0, None
    for M in ins:
        if not M.is_mfma:
            continue
        n_mfma += 1
        # walk forward: (index, wait states elapsed since M issued, wait states the matrix pipe is still busy with the last MFMA seen)
        stack, seen = [(M.idx + 1, M.wait, M.passes)], set()
        while stack:
            i, t, busy = stack.pop()
            while i < len(ins) and t < required:
                J = ins[i]
                if (i, t) in seen:
                    break
                seen.add((i, t))
                if J.is_mfma:
                    t = max(t, busy)                       # in-order matrix pipe: issues when the previous MFMA has left it
                    dep_c = J.c and J.c == M.dst and J.dst == M.dst
                    hit = (J.ab & M.dst) or ((J.c & M.dst) and not dep_c)
                    if hit and t < required:
                        bad.append((M, J, t))
                    if hit:
                        min_seen = t if min_seen is None else min(min_seen, t)
                    if J.dst & M.dst and not hit:
                        break                               # the registers now belong to J's result: J's own walk covers them
                    busy = t + J.passes
                    t += 1
                else:
                    hit = J.touch & M.dst
                    if hit:
                        if t < required:
                            bad.append((M, J, t))
                        min_seen = t if min_seen is None else min(min_seen, t)
                    t += J.wait
                    if J.kind == "end":
                        break
                    if J.kind == "branch":                  # s_branch: only the target
                        i = labels.get(J.target, len(ins))
                        continue
                    if J.kind == "cbranch" and J.target in labels:
                        stack.append((labels[J.target], t, busy))
                i += 1
    return bad, {"mfma": n_mfma, "instructions": len(ins), "closest_touch_wait_states": min_seen}


DOT_REQUIRED = 3


def audit_dots(asm_text, kernel=None, required=DOT_REQUIRED):
    """The other hazard an inline-asm statement can hide from hipcc (GCNHazardRecognizer's DotWriteDifferentVALURead / -Write on
    gfx940+): the result of a v_dot* instruction may be touched by an instruction of ANOTHER opcode only `required` wait states
    later; the same opcode accumulating into it (SrcC == vDst) needs none.  dg_small.hip's squared norms are asm `v_dot2c_f32_bf16`
    (the builtin was miscompiled): the compiler pads its own dot instructions, not these.
    -> (violations as (dot Ins, offending Ins, wait states), number of dot instructions seen)"""
    ins, labels = parse(asm_text, kernel)
    bad, ndot = [], 0
    for M in ins:
        if not M.mnem.startswith("v_dot"):
            continue
        ndot += 1
        dst = regs_of(M.ops[0]) if M.ops else set()
        stack, seen = [(M.idx + 1, 1)], set()
        while stack:
            i, t = stack.pop()
            while i < len(ins) and t < required:
                J = ins[i]
                if (i, t) in seen:
                    break
                seen.add((i, t))
                touch = J.touch if not J.is_mfma else (J.dst | J.ab | J.c)
                if touch & dst:
                    if J.mnem == M.mnem and regs_of(J.ops[0]) == dst:
                        break                               # the same dot accumulating on: its own walk covers the registers
                    bad.append((M, J, t))
                t += J.wait
                if J.kind == "end":
                    break
                if J.kind == "branch":
                    i = labels.get(J.target, len(ins))
                    continue
                if J.kind == "cbranch" and J.target in labels:
                    stack.append((labels[J.target], t))
                i += 1
    return bad, ndot


def main():
    text = open(sys.argv[1]).read()
    bad, stats = audit(text, sys.argv[2] if len(sys.argv) > 2 else None)
    print(stats)
    for M, J, t in bad[:40]:
        print(f"line {J.line_no}: `{J.text}` touches the result of line {M.line_no} `{M.text}` after {t} wait states (< {REQUIRED})")
    print(f"{len(bad)} violation(s)")
    dbad, ndot = audit_dots(text, sys.argv[2] if len(sys.argv) > 2 else None)
    for M, J, t in dbad[:40]:
        print(f"line {J.line_no}: `{J.text}` touches the result of line {M.line_no} `{M.text}` after {t} wait states (< {DOT_REQUIRED})")
    print(f"{ndot} dot instruction(s), {len(dbad)} violation(s)")
    return 1 if (bad or dbad) else 0


if __name__ == "__main__":
    sys.exit(main())
