#!/usr/bin/env python3
"""Static instruction counts of one kernel by source-line range, from `hipcc -S -gline-tables-only` output.
usage: isa_phase_counts.py file.s mangled-substring name:first:last [name:first:last ...]
Instructions inlined from other files (headers) are attributed by line number alone, so pass ranges of the kernel's own file only."""
import re, sys
from collections import Counter
path, sym, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
phases = [(n, int(a), int(b)) for n, a, b in (s.split(":") for s in specs)]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sym) + r"\S*:", l)][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def ph(line):
    if line is None: return "prologue"
    for n, a, b in phases:
        if a <= line <= b: return n
    return "other:%d" % line
cur = None
valu, lds, salu, vmem, tot = Counter(), Counter(), Counter(), Counter(), Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    s = l.strip()
    if not s or s[0] in ".;" or s.endswith(":"): continue
    op, p = s.split()[0], ph(cur)
    tot[p] += 1
    if op.startswith("v_"): valu[p] += 1
    elif op.startswith("ds_"): lds[p] += 1
    elif op.startswith("s_"): salu[p] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): vmem[p] += 1
for p in sorted(tot, key=lambda k: -tot[k]):
    print("%-18s total %5d  valu %5d  lds %4d  salu %4d  vmem %3d" % (p, tot[p], valu[p], lds[p], salu[p], vmem[p]))
print("all: total %d valu %d lds %d salu %d" % (sum(tot.values()), sum(valu.values()), sum(lds.values()), sum(salu.values())))
