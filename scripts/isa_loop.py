#!/usr/bin/env python3
"""Print the main loop (largest backward branch) of a kernel from hipcc's --save-temps assembly, comments stripped.
usage: isa_loop.py <file.s> <substring of the mangled kernel name> [first line [last line]]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w.$]*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = [l for l in lines[start:end + 1]]
labels, instrs = {}, []
for l in body:
    s = l.strip()
    if not s or s.startswith(";") and not s.startswith(";;#ASM"):
        continue
    m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
    if m:
        labels[m.group(1)] = len(instrs)
        instrs.append(s.split(";")[0].strip())
        continue
    if s.startswith(".") or s.startswith(";;#ASM"):
        continue
    instrs.append(s.split(";")[0].rstrip())
best = None
for k, s in enumerate(instrs):
    if s.startswith(("s_cbranch", "s_branch")):
        tgt = s.split()[-1]
        if tgt in labels and labels[tgt] <= k and (best is None or k - labels[tgt] > best[0]):
            best = (k - labels[tgt], labels[tgt], k)
a, b = (best[1], best[2]) if best else (0, len(instrs) - 1)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
for n, s in enumerate(instrs[a:b + 1]):
    if lo <= n <= hi:
        print(f"{n:5d}  {s}")
