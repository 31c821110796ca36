#!/usr/bin/env python3
"""Scan a gfx950 assembly listing (hipcc -S --cuda-device-only) for a hazard the compiler cannot see inside inline asm:

    a VALU instruction that writes an SGPR (v_readfirstlane_b32, v_readlane_b32, v_cmp* with an SGPR destination)
    followed within five wait states by a VECTOR-MEMORY instruction *inside an inline-asm block* that reads that SGPR
    (the saddr of global_load_* / global_load_lds_*, or m0).

The hazard recogniser pads this with s_nop for instructions it emitted itself, but an asm block is opaque to it, so the
load would go out with the SGPR's previous contents (seen as a memory-aperture fault in round 2).  Exit status 1 and one
line per finding; used by tests/test_host_logic.py on the kernels that issue their own loads.
"""
import re
import sys

SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
VALU_SGPR_WRITE = re.compile(r"^\s*(v_readfirstlane_b32|v_readlane_b32)\s+s(\d+)\b")
VCMP_SGPR = re.compile(r"^\s*v_cmpx?_\w+\s+s\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")
NEED = 5


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(path):
    findings = []
    kernel = "?"
    recent = []                      # (wait states ago, sgpr, line no, text)
    in_asm = False
    for no, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(":") and not s.startswith("."):
            kernel = s[:-1]
            recent = []
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        if in_asm and VMEM.match(s):
            used = sregs(s.split(None, 1)[1] if " " in s else "")
            for ago, reg, wno, wtext in recent:
                if reg in used and ago < NEED:
                    findings.append(f"{path}:{no}: {kernel}: '{s}' reads s{reg} {ago} wait state(s) after "
                                    f"'{wtext}' (line {wno}); needs {NEED}")
        states = 1
        m = re.match(r"^\s*s_nop\s+(\d+)", s)
        if m:
            states = int(m.group(1)) + 1
        recent = [(ago + states, reg, wno, wtext) for ago, reg, wno, wtext in recent if ago + states < NEED + 1]
        m = VALU_SGPR_WRITE.match(s)
        if m:
            recent.append((0, int(m.group(2)), no, s))
        m = VCMP_SGPR.match(s)
        if m:
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                recent.append((0, r, no, s))
    return findings


if __name__ == "__main__":
    bad = [f for p in sys.argv[1:] for f in scan(p)]
    print("\n".join(bad) if bad else "no VALU-SGPR -> asm VMEM hazards")
    sys.exit(1 if bad else 0)
