#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot budget), kernel-trace only;
# gfx950 correction: FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B -> double it.
# Usage (on the GPU box): bash scripts/collect_pmc_traffic.sh <round-tag>     -> profiles/<tag>_pmc_traffic.json
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic_$TAG
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
mx = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mlp_" in k or "segment_reduce" in k:
                name = k.split("(")[1].split("::")[-1] if k.startswith("void (") else k
                name = k[k.index("::") + 2:].split("(")[0] if "::" in k else k
                agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
                mx[name][r["Counter_Name"]] = max(mx[name][r["Counter_Name"]], float(r["Counter_Value"]))
res = {}
for name, v in agg.items():
    n = cnt[name]["FETCH_SIZE"]
    fetch_kb, write_kb = v["FETCH_SIZE"] / max(n, 1), v["WRITE_SIZE"] / max(cnt[name]["WRITE_SIZE"], 1)
    res[name] = {"dispatches": n, "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
                 "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
                 # the largest dispatches of a kernel (e.g. the level-1 aggregation among all segment reductions)
                 "hbm_bytes_largest_launch": (2.0 * mx[name]["FETCH_SIZE"] + mx[name]["WRITE_SIZE"]) * 1024.0}
json.dump({"how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 1 --no-cpu-baseline` (hipGraph steps + the instrumented eager pass + the standalone level-1 aggregation); "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE halves wide coalesced reads)", "kernels": res},
          open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
