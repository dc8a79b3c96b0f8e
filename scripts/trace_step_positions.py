"""Per-position kernel durations of a hipGraph-replayed rollout step from a rocprofv3 --kernel-trace CSV: the trace is cut at every
rollout_advance_kernel (the last launch of a step), the steady-state steps (same launch count) are aligned, and the mean / min duration
of every position is printed with its kernel — the in-graph counterpart of scripts/step_breakdown.py's eager event timings.
usage: trace_step_positions.py <dir with *_kernel_trace.csv>"""
import csv, glob, re, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(void )?([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    return ((m.group(2) + (m.group(3) or "")) if m else n)[:60]
steps, cur = [], []
for r in rows:
    cur.append((short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if "rollout_advance" in r["Kernel_Name"]:
        steps.append(cur); cur = []
lens = collections.Counter(len(s) for s in steps)
L = lens.most_common(1)[0][0]
good = [s for s in steps if len(s) == L][2:]
print(f"{len(steps)} steps in the trace, {len(good)} steady-state steps of {L} launches; step period "
      f"{statistics.median((b[-1][3] - a[-1][3]) / 1e3 for a, b in zip(good, good[1:])):.1f} us (median, end to end)")
tot = 0.0
for i in range(L):
    d = [s[i][1] for s in good]
    tot += statistics.mean(d)
    print(f"{i:3d} {good[0][i][0]:62s} mean {statistics.mean(d):8.1f} us  min {min(d):8.1f}")
print(f"sum of mean kernel durations {tot:.1f} us")
