#!/usr/bin/env python3
"""Basic-block skeleton of a kernel from hipcc's assembly: per block the counts of vector / matrix / memory / LDS / scalar instructions
and where it branches — to find the heavy blocks of a long kernel without reading 8 000 lines.
usage: isa_blocks.py <file.s> <substring of the mangled kernel name> [min instructions per printed block]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w.$]*:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = [], {"label": "entry", "ins": []}
for l in lines[start + 1:end + 1]:
    s = l.strip()
    m = re.match(r"^(\.LBB[\w.$]*):", s)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "ins": []}; continue
    if not s or s.startswith(";") or s.startswith("."): continue
    cur["ins"].append(s.split(";")[0].strip())
blocks.append(cur)
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
for b in blocks:
    c = {}
    for i in b["ins"]:
        k = kind(i.split()[0]); c[k] = c.get(k, 0) + 1
    br = [i for i in b["ins"] if i.startswith(("s_cbranch", "s_branch"))]
    if len(b["ins"]) >= minn or c.get("barrier"):
        print(f"{b['label']:12s} n={len(b['ins']):4d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + ("   -> " + ", ".join(x.split()[0][2:] + " " + x.split()[-1] for x in br) if br else ""))
