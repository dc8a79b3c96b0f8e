#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC slot budget), kernel-trace only;
# gfx950 correction: FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B -> double it.
# Two pass pairs: (A) the bench with its roofline block (per-kernel bytes per launch, incl. the standalone level-1 aggregation),
# (B) the bench without it (only rollout steps run: total bytes / steps executed = HBM traffic per rollout step).
# Usage (on the GPU box): bash scripts/collect_pmc_traffic.sh <round-tag> [workload]   -> profiles/<tag>_pmc_traffic[_<workload>].json
set -e
TAG=${1:-r04}
WL=${2:-headline}
SFX=""; [ "$WL" != "headline" ] && SFX="_$WL"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic_$TAG$SFX
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/A_$c -o p -- python bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-strict-range --no-side-configs > $OUT/A_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/B_$c -o p -- python bench.py --workload $WL --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --no-strict-range --no-side-configs > $OUT/B_$c.log 2>&1
done
python - "$OUT" "$TAG" "$SFX" "$WL" <<'PY'
import csv, glob, collections, json, sys
out, tag, sfx, wl = sys.argv[1:5]
def short(k):
    return k[k.index("::") + 2:].split("(")[0] if "::" in k else k.split("(")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
mx = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/A_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mlp_" in k or "segment_reduce" in k:
                name = short(k)
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
# (B) whole rollout steps: every dispatch of the process, divided by the steps executed (= dispatches of rollout_advance_kernel)
tot = collections.defaultdict(float); per_kernel = collections.defaultdict(lambda: collections.defaultdict(float)); steps = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    n_adv = 0
    for f in glob.glob(f"{out}/B_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot[c] += float(r["Counter_Value"]); per_kernel[short(r["Kernel_Name"])][c] += float(r["Counter_Value"])
            n_adv += "rollout_advance" in r["Kernel_Name"]
    steps[c] = n_adv
per_step = None
if steps.get("FETCH_SIZE") and steps.get("WRITE_SIZE"):
    per_step = {"steps_executed": steps["FETCH_SIZE"],
                "hbm_bytes_per_step": (2.0 * tot["FETCH_SIZE"] / steps["FETCH_SIZE"] + tot["WRITE_SIZE"] / steps["WRITE_SIZE"]) * 1024.0,
                "read_bytes_per_step": 2.0 * tot["FETCH_SIZE"] / steps["FETCH_SIZE"] * 1024.0,
                "write_bytes_per_step": tot["WRITE_SIZE"] / steps["WRITE_SIZE"] * 1024.0,
                "by_kernel_bytes_per_step": {k: (2.0 * v["FETCH_SIZE"] / steps["FETCH_SIZE"] + v["WRITE_SIZE"] / steps["WRITE_SIZE"]) * 1024.0
                                             for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["FETCH_SIZE"])[:12]},
                "note": "includes the one-off first step (plans, weight packing) spread over the steps executed"}
json.dump({"workload": wl,
           "how": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --workload {wl} --steps 4 --warmup 1 --no-cpu-baseline` (hipGraph steps + the instrumented eager pass + the standalone level-1 aggregation); "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE halves wide coalesced reads); `per_step`: the same two passes over "
                  "the bench with --no-roofline (only rollout steps run), all dispatches / steps executed",
           "kernels": res, "per_step": per_step},
          open(f"profiles/{tag}_pmc_traffic{sfx}.json", "w"), indent=1)
print(json.dumps({"kernels": res, "per_step": per_step}, indent=1))
PY
