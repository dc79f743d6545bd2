#!/usr/bin/env python3
"""Static check of hipcc's device assembly for a gfx940/gfx950 hazard the compiler does not cover for INLINE ASM operands.

On gfx940+ a VALU instruction that writes an SGPR (v_readlane / v_readfirstlane -- i.e. every restore of a spilled scalar
register --, v_cmp with a scalar destination, carry-outs) must be followed by 2 wait states before a VALU instruction reads
that SGPR as a constant, and by 5 before a VMEM instruction uses it as an address (ISA "required software-inserted wait
states").  hipcc's hazard recogniser inserts them between its own instructions; an inline asm statement is opaque to it
-- and this library's hot loops read their polynomial constants ("s"(C) operands of fm::fma_sc / horner2x*) and their row
pointers (global_store ... s[base]) from SGPRs inside asm statements.  Seen in round 4: with four more live scalars the GBM
generator spilled a Horner constant, the restore landed right before the asm FMA, the FMA read the register's PREVIOUS
content and the paths came out 1e-10 off (tests/test_gpu_parity.py caught it at its 1e-11).

A second hazard of the same family, and the one that actually bit (round 4, found by comparing the two builds' matrices
element by element, tools/diag_stamps.py): on gfx940+ a VMEM STORE of more than 64 bits must be followed by 2 wait states
before a VALU instruction overwrites the store's DATA registers -- the memory pipeline reads them after issue.  For its own
stores hipcc inserts the wait states; `global_store_dwordx4` inside an asm statement is opaque to it.  In the failing
build the generator's next instruction overwrote half of the data pair: rows 0, 1 and 5 of the matrix held values with a
foreign low or high word in a quarter of the lanes (relative error 2^-33 = 1.2e-10, or percent) while the price chain in
the registers stayed right.  The store's asm statement now carries its own `s_nop 1`.

A third member of the family is checked as well: the result of a transcendental VALU instruction read by an asm VALU
instruction in the very next wait state (gfx940+ forwarding hazard).

Usage: check_asm_hazards.py file.s [...]   -> lists every asm statement that reads an SGPR too soon after a VALU write;
exit code 1 if any."""
import re
import sys

VALU_SGPR_WRITERS = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+(s\d+|s\[\d+:\d+\])")
VCMP_SGPR = re.compile(r"^\s*v_cmp\w*_e64\s+(s\[\d+:\d+\])")
CARRY_SGPR = re.compile(r"^\s*v_(?:add|sub|subrev)_co_u32(?:_e64)?\s+v\d+,\s*(s\[\d+:\d+\])")
MADCARRY = re.compile(r"^\s*v_mad_[ui]64_[ui]32\s+v\[\d+:\d+\],\s*(s\[\d+:\d+\])")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in SREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def written_sgprs(line):
    for pat in (VALU_SGPR_WRITERS, VCMP_SGPR, CARRY_SGPR, MADCARRY):
        m = pat.match(line)
        if m:
            return regs(m.group(m.lastindex))
    return set()


def is_insn(line):
    t = line.strip()
    return bool(t) and not t.startswith((";", ".", "//")) and not t.endswith(":")


def wait_states(line):
    m = re.match(r"\s*s_nop\s+(\d+)", line)
    return int(m.group(1)) + 1 if m else 1


VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WIDE_STORE = re.compile(r"^\s*(?:global|flat|buffer|scratch)_store_dwordx[34]\b")


def vregs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_wide_stores(path):
    """asm-embedded stores of > 64 bits whose data VGPRs a VALU instruction writes within the next 2 wait states"""
    bad = []
    lines = open(path, errors="ignore").read().splitlines()
    kernel, in_asm = "?", False
    pending = []   # [data vgprs, wait states since the store, line, text]
    for n, line in enumerate(lines, 1):
        t = line.strip()
        head = t.split(";")[0].strip()
        if head.endswith(":") and head.startswith("_Z"):
            kernel = head[:-1]
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not is_insn(line):
            continue
        body = t.split(";")[0]
        if body.startswith("v_") and not body.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dst = vregs(body.split(None, 1)[1].split(",")[0]) if " " in body else set()
            for data, age, ln, txt in pending:
                if age < 2 and dst & data:
                    bad.append((path, ln, kernel, txt, sorted(dst & data), age, n, body))
        ws = wait_states(line)
        pending = [(d, age + ws, ln, txt) for d, age, ln, txt in pending if age + ws < 2]
        if in_asm and WIDE_STORE.match(body):
            ops = body.split(None, 1)[1].split(",")
            pending.append((vregs(ops[1]), 0, n, body))   # global_store vaddr, vdata, saddr
    return bad


TRANS = re.compile(r"^\s*v_(?:rcp|rsq|sqrt|exp|log|sin|cos)(?:_iflag|_clamp|_legacy)?_f(?:16|32|64)(?:_e32|_e64)?\s+(v\d+|v\[\d+:\d+\])")


def check_trans(path):
    """gfx940+: the result of a transcendental VALU instruction (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) must
    not be read by a non-transcendental VALU instruction in the very next wait state; asm statements that read such a
    register right behind its producer are listed."""
    bad = []
    lines = open(path, errors="ignore").read().splitlines()
    kernel, in_asm, pending = "?", False, None   # pending: (vgprs, line) of a trans op issued in the previous wait state
    for n, line in enumerate(lines, 1):
        t = line.strip()
        head = t.split(";")[0].strip()
        if head.endswith(":") and head.startswith("_Z"):
            kernel = head[:-1]
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not is_insn(line):
            continue
        body = t.split(";")[0]
        if in_asm and pending and body.startswith("v_") and not TRANS.match(body):
            ops = body.split(None, 1)[1] if " " in body else ""
            src = ops.split(",", 1)[1] if "," in ops else ""
            if vregs(src) & pending[0]:
                bad.append((path, n, kernel, body, sorted(vregs(src) & pending[0]), pending[1]))
        m = TRANS.match(body)
        pending = (vregs(m.group(1)), n) if m else None
    return bad


def check(path):
    bad = []
    lines = open(path, errors="ignore").read().splitlines()
    kernel = "?"
    pending = []   # [regs, wait states since the write]
    in_asm = False
    for n, line in enumerate(lines, 1):
        t = line.strip()
        head = t.split(";")[0].strip()
        if head.endswith(":") and head.startswith("_Z"):
            kernel = head[:-1]
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not is_insn(line):
            continue
        if in_asm:
            body = t.split(";")[0]
            ops = body.split(None, 1)[1] if " " in body else ""
            vmem = body.startswith(("global_", "buffer_", "flat_", "scratch_"))
            # sources only: drop the first operand of a VALU instruction (its destination)
            src = ops if vmem else (ops.split(",", 1)[1] if "," in ops else "")
            used = regs(src)
            need = 5 if vmem else 2
            for r, age in pending:
                if age < need and used & r:
                    bad.append((path, n, kernel, t, sorted(used & r), age, need))
        w = written_sgprs(line)
        ws = wait_states(line)
        pending = [(r, age + ws) for r, age in pending if age + ws < 5]
        if w:
            pending.append((w, 0))
    return bad


if __name__ == "__main__":
    allbad, stores, trans = [], [], []
    for f in sys.argv[1:]:
        allbad += check(f)
        stores += check_wide_stores(f)
        trans += check_trans(f)
    for path, n, kernel, t, r, age, need in allbad:
        print(f"{path}:{n}: {kernel}: asm `{t}` reads s{r} {age} wait state(s) after a VALU write (needs {need})")
    for path, ln, kernel, txt, regs_, age, n, body in stores:
        print(f"{path}:{ln}: {kernel}: asm `{txt}`: its data v{regs_} is overwritten {age} wait state(s) later by `{body}` (line {n}; needs 2)")
    for path, n, kernel, body, regs_, ln in trans:
        print(f"{path}:{n}: {kernel}: asm `{body}` reads v{regs_}, the result of the transcendental instruction one line up (line {ln}; needs 1 wait state)")
    print(f"{len(allbad) + len(stores) + len(trans)} hazard(s) in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if allbad or stores or trans else 0)
