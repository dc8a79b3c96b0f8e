#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc's --save-temps assembly.
usage: isa_count.py <file.s> <substring of the mangled kernel name> [--blocks]
Finds the kernel, takes the span of its largest backward branch (the persistent pair loop) and counts
instructions by issue class (MFMA, VALU, transcendental, packed-f32, LDS, VMEM, SALU, waits / barriers)."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_pk_") and "f32" in op:
        return "valu_pk32"
    if op.startswith("v_accvgpr"):
        return "valu_acc_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w.$]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    instrs = []          # (index in body, op, text)
    for i, l in enumerate(body):
        s = l.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        instrs.append((i, op, s))
    # largest backward branch
    best = None
    for k, (i, op, s) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                span = k - labels[tgt]
                if best is None or span > best[0]:
                    best = (span, labels[tgt], k)
    print(f"kernel instructions: {len(instrs)}")
    tot = Counter(classify(op) for _, op, _ in instrs)
    print("whole kernel:", dict(tot))
    if best is None:
        return
    span, a, b = best
    loop = instrs[a:b + 1]
    c = Counter(classify(op) for _, op, _ in loop)
    print(f"main loop: {len(loop)} instructions")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"  {k:14s} {v}")
    nv = c["valu"] + c["valu_trans"] + c["valu_pk32"] + c["valu_acc_mov"]
    print(f"  VALU (all kinds) per MFMA: {nv / max(c['mfma'], 1):.2f}")
    ops = Counter(op for _, op, _ in loop if classify(op).startswith("valu"))
    print("  VALU opcodes:")
    for k, v in ops.most_common(60):
        print(f"    {k:28s} {v}")
    if "--blocks" in sys.argv:
        inv = {}
        for name, idx in labels.items():
            inv.setdefault(idx, []).append(name)
        cur, curname = Counter(), "(loop head)"
        def flush():
            if sum(cur.values()):
                v = cur["valu"] + cur["valu_trans"] + cur["valu_pk32"] + cur["valu_acc_mov"]
                print(f"  block {curname:12s} n={sum(cur.values()):4d} mfma={cur['mfma']:3d} valu={v:4d} trans={cur['valu_trans']:3d} lds={cur['lds']:3d} vmem={cur['vmem']:3d} salu={cur['salu']:3d} wait={cur['waitcnt']:3d} bar={cur['barrier']} last={last}")
        last = ""
        for k in range(a, b + 1):
            if k in inv and k != a:
                flush(); cur = Counter(); curname = inv[k][0]
            cur[classify(instrs[k][1])] += 1
            last = instrs[k][2][:40]
            if instrs[k][1].startswith(("s_cbranch", "s_branch")):
                pass
        flush()
    if "--spills" in sys.argv:
        print("  scratch:", sum(1 for _, op, _ in loop if op.startswith("scratch_")))


if __name__ == "__main__":
    main()
