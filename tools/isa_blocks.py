#!/usr/bin/env python3
"""instruction mix of the largest basic blocks of one kernel in a hipcc -S listing:  tools/isa_blocks.py tse.s k_remapILi1"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if pat in l and l.rstrip().split(";")[0].strip().endswith(":") and l.startswith("_Z"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = []; cur = ("entry", [])
for l in lines[start + 1:end]:
    s = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", s):
        blocks.append(cur); cur = (s, [])
    elif s and not s.startswith(";") and not s.startswith("."):
        cur[1].append(s)
blocks.append(cur)
print("total instructions", sum(len(b[1]) for b in blocks), "blocks", len(blocks))
for name, ins in sorted(blocks, key=lambda b: -len(b[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 5]:
    c = collections.Counter(i.split()[0] for i in ins)
    cats = collections.Counter()
    for op, n in c.items():
        if "dpp" in op: cats["dpp"] += n
        elif op.startswith("ds_"): cats["lds"] += n
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): cats["vmem_" + ("st" if "store" in op else "ld") + ("_scratch" if op.startswith("scratch") else "")] += n
        elif op.startswith("s_waitcnt"): cats["waitcnt"] += n
        elif op.startswith("s_"): cats["salu"] += n
        elif op.startswith("v_") and "f64" in op: cats["f64_" + ("rcp/div" if ("rcp" in op or "div_" in op) else ("fma" if "fma" in op else "other"))] += n
        elif op.startswith("v_"): cats["valu32"] += n
        else: cats["other"] += n
    print(name, len(ins), dict(cats))
    print("    ", c.most_common(16))
