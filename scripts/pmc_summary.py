"""Summarise a rocprofv3 --pmc csv: per kernel, counters summed over dispatches + dispatch count and time."""
import csv, glob, collections, sys
d = sys.argv[1]
dur = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; cnt[k] += 1
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "g4c" in k or "mlp_" in k or "segment_reduce" in k:
            print(k, f"dispatches={cnt[k]} total_us={dur[k]:.1f}")
            for c, x in sorted(v.items()):
                print(f"    {c:32s} {x:16.0f}  per-dispatch {x / max(cnt[k], 1):14.0f}")
