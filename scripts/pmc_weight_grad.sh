#!/bin/bash
# PMC passes over the weight-gradient kernel alone (600k x 128 x 128): utilisation and LDS counters, one small pass each.
# Usage (GPU box): bash scripts/pmc_weight_grad.sh  ->  gpurun_out/pmc_weight_grad.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/wg_run.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from graphs4cfd_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); rows = 600000
g = torch.randn(rows, 128, device=dev); a = torch.randn(rows, 128, device=dev)
scratch = torch.empty(int(lib.g4c_weight_grad_scratch_floats(rows)), device=dev); out = torch.empty(128 * 128 + 128, device=dev)
for _ in range(6):
    lib.g4c_weight_grad(_lib.ptr(g), 128, _lib.ptr(a), 128, rows, _lib.ptr(scratch), _lib.ptr(out), 1, _lib.stream_handle(dev))
torch.cuda.synchronize()
PY
declare -A C
C[util]="MfmaUtil VALUBusy MemUnitStalled"
C[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
C[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
: > gpurun_out/pmc_weight_grad.txt
for p in util lds sq; do
  OUT=gpurun_out/pmc_wg_$p; rm -rf $OUT
  timeout -k 10 240 rocprofv3 --pmc ${C[$p]} --kernel-trace --output-format csv -d $OUT -o p -- python /tmp/wg_run.py > $OUT.log 2>&1
  echo "== $p" >> gpurun_out/pmc_weight_grad.txt
  python - $OUT >> gpurun_out/pmc_weight_grad.txt <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]; dur = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; cnt[k] += 1
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "weight_grad" in k or "colsum" in k:
            print(k, f"dispatches={cnt[k]} avg_us={dur[k] / max(cnt[k], 1):.1f}")
            for c, x in sorted(v.items()):
                print(f"    {c:32s} per-dispatch {x / max(cnt[k], 1):16.1f}")
PY
done
cat gpurun_out/pmc_weight_grad.txt
