"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (hipGraph-replayed rollout steps): per kernel name the
launches, total duration, and the gap in front of it; and the totals.  usage: trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2          # the second half: steady-state replays
rows = rows[skip:]
dur = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.Counter()
prev_end = None; tot_d = tot_g = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"([A-Za-z_][A-Za-z_0-9]*)\s*(<[^()]*>)?\s*\([^()]*\)\s*(\[clone[^]]*\])?\s*$", r["Kernel_Name"].replace("(anonymous namespace)", "anon"))
    name = (m.group(1) if m else r["Kernel_Name"])[:40]
    d = (e - s) / 1e3; g = max(0.0, (s - prev_end) / 1e3) if prev_end is not None else 0.0
    if g > 200: g = 0.0          # (host-side pauses between replays)
    dur[name] += d; gap[name] += g; cnt[name] += 1; tot_d += d; tot_g += g; prev_end = max(e, prev_end or 0)
print(f"{len(rows)} kernels: busy {tot_d / 1e3:.2f} ms, idle between kernels {tot_g / 1e3:.2f} ms ({100 * tot_g / (tot_d + tot_g):.1f} %), mean gap {tot_g / len(rows):.2f} us")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"  {k:42s} n={cnt[k]:6d}  mean {dur[k] / cnt[k]:7.2f} us  mean gap in front {gap[k] / cnt[k]:5.2f} us")
