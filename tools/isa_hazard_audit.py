#!/usr/bin/env python3
"""Audit a gfx950 assembly listing (hipcc -save-temps: *-hip-amdgcn-amd-amdhsa-gfx950.s) for MFMA results touched too early.

hipcc pads nothing around an inline-asm MFMA and does not order plain register arithmetic against a bare
`asm volatile("s_nop ..." ::: "memory")`: a compiler-placed VALU read of a score register can end up directly behind the
MFMA that writes it (round 5: the reference maxima of the four-wave attention forward -- valid results, but differing in
the last bit between identical calls under contention).  This walks the instruction stream behind every v_mfma and
reports any instruction that reads or writes a register of its destination before the required wait states have passed
(8-pass product: 12 states, 4-pass: 8; cdna_hip_programming.md section 5.7 item 2), except the next MFMA taking the
destination whole as its C operand (accumulate chain: 0 states).  States are counted conservatively: one per
instruction, n + 1 per `s_nop n`, PASSES per intervening MFMA (the matrix pipe takes one product of a wave at a time).

Two more rules of the same kind are checked: a vector write fewer than 2 wait states ahead of an MFMA reading it as an
operand, and a wide buffer store with an SGPR soffset whose data registers are overwritten within 2 wait states.

Rule 1 follows branches (a product at the end of a loop body is read at the loop's top) and reports pairs with an inline-asm
instruction on either side; `--all` adds the pairs of two compiler-generated instructions (hipcc pads those itself and, across
taken branches, relies on the refetch time that a count of instructions does not see).

  python tools/isa_hazard_audit.py file.s [kernel-name-substring] [--all]      exit code 1 if anything is reported
"""
import re
import sys

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def passes(op):
    return 8 if "32x32" in op else 4


def audit(path, only=None, everything=False):
    lines = open(path).read().split("\n")
    found = []
    fn = None
    insts = []   # (function, line number, text)
    in_asm = set()  # line numbers of instructions inside an inline-asm statement
    labels = {}  # (function, label) -> index of the first instruction behind it
    asm = False
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
        if ";;#ASMSTART" in l:
            asm = True
        elif ";;#ASMEND" in l:
            asm = False
        elif asm:
            in_asm.add(i + 1)
        t = l.split(";")[0].strip()
        if not t or fn is None:
            continue
        if t.endswith(":"):
            labels[(fn, t[:-1])] = len(insts)
            continue
        if t.startswith("."):
            continue
        insts.append((fn, i + 1, t))
    for k, (fn, ln, t) in enumerate(insts):
        if not t.startswith("v_mfma") or (only and only not in fn):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        dst = regs(ops[0])
        need = 12 if passes(t) == 8 else 8
        # every path behind the MFMA, branches followed (a product at the end of a loop body is read at the loop's top)
        work, seen = [(k + 1, 0)], set()
        while work:
            j, states = work.pop()
            while j < len(insts) and states < need and (j, states) not in seen:
                seen.add((j, states))
                fn2, ln2, t2 = insts[j]
                if fn2 != fn:
                    break
                op = t2.split()[0]
                if op in ("s_endpgm", "s_setpc_b64"):
                    break
                if op == "s_branch" or op.startswith("s_cbranch"):
                    target = labels.get((fn, t2.split()[1]))
                    if target is not None:
                        work.append((target, states + 1))
                    if op == "s_branch":
                        break
                    states += 1
                    j += 1
                    continue
                if op == "s_nop":
                    states += int(t2.split()[1]) + 1
                    j += 1
                    continue
                if op.startswith("v_mfma"):
                    o2 = [o.strip() for o in t2.split(None, 1)[1].split(",")]
                    chain = regs(o2[0]) == dst and regs(o2[3]) == dst and not (regs(o2[1]) | regs(o2[2])) & dst
                    if not chain and regs(t2) & dst:
                        found.append((fn, ln, t, ln2, t2, states))
                    if chain:
                        break  # the chain's next link is audited on its own
                    states += passes(t2)
                    j += 1
                    continue
                if op.startswith(("v_", "ds_", "buffer_", "global_", "flat_", "scratch_")) and regs(t2) & dst:
                    found.append((fn, ln, t, ln2, t2, states))
                    break
                states += 1
                j += 1
    # Pairs of two compiler-generated instructions are the compiler's business (its hazard recognizer pads them; across
    # taken branches it relies on the refetch time, which this count of states does not see): only pairs with an inline-asm
    # instruction on either side are reported unless everything is asked for.
    if not everything:
        found = [f for f in found if f[1] in in_asm or f[3] in in_asm]
    return sorted(set(found), key=lambda f: (f[1], f[3]))


def audit_operands(path, only=None):
    """Second rule of the same section: a VALU (or v_accvgpr_write) result needs 2 wait states before an MFMA reads it as an
    A / B / C operand.  Reports MFMAs whose operand register was written by a vector instruction fewer than 2 states earlier."""
    lines = open(path).read().split("\n")
    found = []
    fn = None
    window = []  # the last instructions of the current function: (line number, text, states it accounts for)
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
            window = []
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or fn is None:
            continue
        if t.endswith(":"):
            window = []  # a label: predecessors unknown, the straight-line rule ends here
            continue
        op = t.split()[0]
        if op.startswith("v_mfma") and not (only and only not in fn):
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            src = regs(ops[1]) | regs(ops[2]) | regs(ops[3])
            states = 0
            for ln2, t2, st2 in reversed(window):
                if states >= 2:
                    break
                op2 = t2.split()[0]
                if op2.startswith("v_") and not op2.startswith(("v_mfma", "v_cmp", "v_nop")):
                    dst2 = regs(t2.split(None, 1)[1].split(",")[0])
                    if dst2 & src:
                        found.append((fn, i + 1, t, ln2, t2, states))
                        break
                states += st2
        window.append((i + 1, t, int(t.split()[1]) + 1 if op == "s_nop" else 1))
        del window[:-4]
    return found


def audit_stores(path, only=None):
    """Third rule (observed on gfx950, csrc/gemm.hip epilogue): a buffer store of more than 8 bytes whose data registers
    are overwritten by a vector instruction within the next 2 wait states stores part of the NEW value.  hipcc pads one
    state itself unless the store's soffset is an SGPR, where it assumes no hazard -- those stores are audited here."""
    lines = open(path).read().split("\n")
    found = []
    fn = None
    pending = []  # stores still inside their 2-state window: [line number, text, data registers, states seen]
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
            pending = []
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":") or fn is None:
            continue
        op = t.split()[0]
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_nop")):
            dst = regs(t.split(None, 1)[1].split(",")[0])
            for ln2, t2, data, st in pending:
                if dst & data:
                    found.append((fn, ln2, t2, i + 1, t, st))
        step = int(t.split()[1]) + 1 if op == "s_nop" else 1
        pending = [[a, b, c, st + step] for a, b, c, st in pending if st + step < 2]
        if op in ("buffer_store_dwordx3", "buffer_store_dwordx4") and not (only and only not in fn):
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            soffset = ops[3].split()[0] if len(ops) > 3 else "off"
            if re.fullmatch(r"s\d+", soffset):
                pending.append([i + 1, t, regs(ops[0]), 0])
    return found


if __name__ == "__main__":
    everything = "--all" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--all"]
    sys.argv[1:] = args
    only = sys.argv[2] if len(sys.argv) > 2 else None
    res = audit(sys.argv[1], only, everything)
    for fn, ln, t, ln2, t2, st in res[:40]:
        print(f"{fn[:60]}: line {ln}: {t}\n    touched after {st} wait state(s) by line {ln2}: {t2}")
    print(f"{len(res)} early touches of an MFMA destination in {sys.argv[1]}")
    res2 = audit_operands(sys.argv[1], only)
    for fn, ln, t, ln2, t2, st in res2[:40]:
        print(f"{fn[:60]}: line {ln}: {t}\n    operand written {st} wait state(s) earlier by line {ln2}: {t2}")
    print(f"{len(res2)} MFMA operands written fewer than 2 wait states ahead in {sys.argv[1]}")
    res3 = audit_stores(sys.argv[1], only)
    for fn, ln, t, ln2, t2, st in res3[:40]:
        print(f"{fn[:60]}: line {ln}: {t}\n    data register overwritten {st} wait state(s) later by line {ln2}: {t2}")
    print(f"{len(res3)} wide buffer stores (SGPR soffset) whose data is overwritten within 2 wait states in {sys.argv[1]}")
    sys.exit(1 if res or res2 or res3 else 0)
