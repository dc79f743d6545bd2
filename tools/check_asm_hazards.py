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


def check(path):
    bad = []
    lines = open(path, errors="ignore").read().splitlines()
    kernel = "?"
    pending = []   # [regs, wait states since the write]
    in_asm = False
    for n, line in enumerate(lines, 1):
        t = line.strip()
        if t.endswith(":") and t.startswith("_Z"):
            kernel = t[:-1]
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
    allbad = []
    for f in sys.argv[1:]:
        allbad += check(f)
    for path, n, kernel, t, r, age, need in allbad:
        print(f"{path}:{n}: {kernel}: asm `{t}` reads s{r} {age} wait state(s) after a VALU write (needs {need})")
    print(f"{len(allbad)} hazard(s) in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if allbad else 0)
