#!/usr/bin/env python
"""ISA lint (round 5, co-residency race hunt): flags every `s_waitcnt lgkmcnt(n > 0)` issued while BOTH a scalar load and an LDS
op are outstanding (the two return out of order, only lgkmcnt(0) is safe then).  Usage: hipcc <build flags> -S --cuda-device-only
-o gemm.s csrc/sdv_gemm.hip; python tools/isa/lgkm_mixed_wait_lint.py gemm.s.  Result on the shipped tree: 0 flags in every kernel."""
import re, sys
pend = []   # list of ('L'|'S', line)
kern = None
flags = {}
for i, ln in enumerate(open(sys.argv[1]), 1):
    s = ln.split(';')[0].strip()
    if not s: continue
    if s.startswith('_Z') and s.endswith(':'):
        kern = s[:-1]; pend = []; continue
    if re.fullmatch(r'\.LBB\d+_\d+:', s): pend = []; continue
    m = s.split()[0]
    if m.startswith(('s_load', 's_buffer_load')): pend.append(('S', i))
    elif m.startswith('ds_'): pend.append(('L', i))
    elif m == 's_waitcnt':
        mm = re.search(r'lgkmcnt\((\d+)\)', s)
        if mm:
            n = int(mm.group(1))
            if n > 0 and any(p[0] == 'S' for p in pend) and any(p[0] == 'L' for p in pend):
                flags.setdefault(kern, []).append((i, n, [p for p in pend][-6:]))
            # retire
            if n == 0: pend = []
            else: pend = pend[-n:]
print(len(flags))
for k, v in flags.items():
    print(k[-60:], len(v), v[:3])
