#!/usr/bin/env python
"""ISA lint (round 5, co-residency race hunt): per kernel, how many DS writes have one of their data / address VGPRs overwritten
within the next N instructions (argv[2], default 2).  Result on the shipped tree: the very next slot overwrites them in EVERY
kernel, the healthy ones included (33 - 100 sites each) - DS write data is sampled at issue, unlike the 128-bit buffer store with
an SGPR offset (DESIGN.md, round-2 hazard); this is not where the two-workgroups-per-CU fault comes from."""
import re, sys
from collections import Counter
def vregs(op):
    op = op.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()
def written(s):
    mnem, _, rest = s.partition(" ")
    if not mnem.startswith(("v_", "ds_read", "buffer_load", "global_load", "scratch_load")) or mnem.startswith("v_cmp"):
        return set()
    if mnem.startswith("buffer_load") and s.endswith("lds"): return set()
    return vregs(rest.split(",")[0])
WIN = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kern = None; lines = []
res = Counter(); tot = Counter(); ex = {}
def flush():
    for i, s in enumerate(lines):
        if s.startswith(("ds_write", "ds_add", "ds_store")):
            ops = s.partition(" ")[2].split(",")
            regs = set()
            for o in ops[:3]: regs |= vregs(o.split()[0] if o.strip() else "")
            tot[kern] += 1
            for j in range(i + 1, min(i + 1 + WIN, len(lines))):
                if lines[j].endswith(":") : break
                w = written(lines[j]) & regs
                if w:
                    res[kern] += 1; ex.setdefault(kern, (s, lines[j], j - i)); break
for ln in open(sys.argv[1]):
    s = ln.split(';')[0].strip()
    if not s or s.startswith('.') and not s.startswith('.LBB'): continue
    if s.startswith('_Z') and s.endswith(':'):
        flush(); kern = s[:-1]; lines = []; continue
    lines.append(s)
flush()
for k in tot:
    m = re.search(r"igemm_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi64ELb(\d)ELi2ELi(\d)E", k)
    tag = "WM%s WN%s TM%s TN%s conv%s FEAT%s" % m.groups() if m else k[-40:]
    print(f"{tag:40s} ds_writes {tot[k]:5d}  overwritten within {WIN}: {res[k]:4d}  {ex.get(k, '')}")
