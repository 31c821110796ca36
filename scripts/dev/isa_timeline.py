#!/usr/bin/env python3
"""Developer aid: run-length timeline of instruction classes of one kernel's ISA (M mfma, v valu, a accvgpr, d ds_read,
D lds-dma, g global load, S store, B barrier, w waitcnt, n nop, s salu, b branch, X scratch)."""
import itertools
import sys
L = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith((';', '.'))]
def cls(l):
    for p, c in (('v_mfma', 'M'), ('ds_read', 'd'), ('global_load_lds', 'D'), ('global_load', 'g'), ('global_store', 'S'),
                 ('s_barrier', 'B'), ('s_waitcnt', 'w'), ('s_nop', 'n'), ('v_accvgpr', 'a'), ('v_', 'v'), ('s_cbranch', 'b'),
                 ('s_branch', 'b'), ('s_', 's'), ('scratch', 'X')):
        if l.startswith(p):
            return c
    return '?'
s = ''.join(cls(l) for l in L)
i = s.find('M')
out = []
for k, g in itertools.groupby(s[i:]):
    n = len(list(g))
    out.append(f"{k}{n if n > 1 else ''}")
print(len(s), ' '.join(out)[:int(sys.argv[2]) if len(sys.argv) > 2 else 8000])
