#!/usr/bin/env python3
"""Developer aid: per-kernel instruction-class counts of a hipcc -S dump (python scripts/dev/isa_stats.py file.s [name-filter])."""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"^_Z\w+:", l)]
for i, name in starts:
    if flt not in name:
        continue
    j = i
    while j < len(L) and "s_endpgm" not in L[j]:
        j += 1
    lines = [l.strip() for l in L[i + 1:j] if l.strip() and not l.strip().startswith((";", ".", "_Z"))]

    def cnt(p):
        return sum(1 for l in lines if re.match(p, l))
    print(name[:70], "| instrs", len(lines), "mfma", cnt(r"v_mfma"), "accread", cnt(r"v_accvgpr_read"), "accwrite",
          cnt(r"v_accvgpr_write"), "scratch", cnt(r"scratch_"), "ds_read", cnt(r"ds_read"), "valu",
          cnt(r"v_(?!mfma|accvgpr)"), "salu", cnt(r"s_"), "v_mov", cnt(r"v_mov_b32"), "gload", cnt(r"global_load"),
          "waitcnt", cnt(r"s_waitcnt"), "nop", cnt(r"s_nop"))
