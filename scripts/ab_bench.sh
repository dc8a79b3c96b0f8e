#!/bin/bash
# Same-box A/B of whole-rollout rates: scripts/ab_bench.sh <workload> <steps> <lib name or "shipped">...   (three interleaved rounds)
cd "$GRAFT_REPO_ROOT"
W=$1; S=$2; shift 2
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = shipped ]; then L=graphs4cfd_amd/lib/libg4c.so; else L=graphs4cfd_amd/lib/libg4c_ws_$v.so; fi
  G4C_LIB_PATH=$PWD/$L timeout 600 python bench.py --workload $W --steps $S --no-cpu-baseline --no-side-configs --no-roofline --no-strict-range 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '$v', round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms')"
done; done
