#!/bin/bash
# Same-box A/B of builds of the weight-stationary kernel (scripts/build_ws_variant.sh <name> -D...): three interleaved rounds of
# scripts/ws_time_one.py per build.  Usage (GPU box): bash scripts/ws_ab_variants.sh shipped poly prio ...   [ROWS=600000]
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do for v in "$@"; do
  if [ "$v" = shipped ]; then L=graphs4cfd_amd/lib/libg4c.so; else L=graphs4cfd_amd/lib/libg4c_ws_$v.so; fi
  G4C_LIB_PATH=$PWD/$L timeout 300 python scripts/ws_time_one.py --rows ${ROWS:-600000} 2>&1 | grep -v amdgpu.ids | tail -1
done; done
