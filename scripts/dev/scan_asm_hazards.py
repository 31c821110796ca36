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


# ---- second hazard: wide buffer store with a REGISTER in the scalar-offset field, data registers overwritten right behind it
# hipcc pads "VMEM store of more than 8 bytes followed by a VALU write of its data registers" with a wait state only when
# the store's soffset field is NOT a register (GCNHazardRecognizer::createsVALUHazard).  On gfx950 the hazard is there with a
# register too: a v_pk_add_f32 into the data registers issued right behind `buffer_store_dwordx4 v[12:15], v135, s[8:11], s19
# offen` changed what was stored (round 3, planned aggregation: element 1 of lanes 12-15 of every 16).
WIDE_STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*v_\w+\s+v(?:\[(\d+):(\d+)\]|(\d+))")


def scan_store_hazard(path):
    findings = []
    kernel = "?"
    pending = None                   # (line no, text, data registers) of a store still within one wait state
    for no, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(":") and not s.startswith("."):
            kernel = s[:-1]
            pending = None
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        if pending is not None:
            m = VDST.match(s)
            if m and not s.startswith("v_mfma"):
                dst = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(3) is None else {int(m.group(3))}
                if dst & pending[2]:
                    findings.append(f"{path}:{no}: {kernel}: '{s}' overwrites the data of '{pending[1]}' (line {pending[0]}) "
                                    f"without a wait state (soffset is a register: hipcc does not pad)")
            pending = None
        m = WIDE_STORE.match(s)
        if m and m.group(4).rstrip(",").startswith("s"):
            pending = (no, s, set(range(int(m.group(2)), int(m.group(3)) + 1)))
    return findings


if __name__ == "__main__":
    bad = [f for p in sys.argv[1:] for f in scan(p) + scan_store_hazard(p)]
    print("\n".join(bad) if bad else "no VALU-SGPR -> asm VMEM hazards, no unpadded wide buffer stores")
    sys.exit(1 if bad else 0)
