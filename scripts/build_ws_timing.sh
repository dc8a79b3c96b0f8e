#!/bin/bash
# libg4c variant whose weight-stationary kernel writes cycle stamps (scripts/ws_stamps.py): graphs4cfd_amd/lib/libg4c_ws_timing.so
set -e
cd "$(dirname "$0")/../graphs4cfd_amd/csrc"
make -j8 >/dev/null      # (NOTE: this also rebuilds ../lib/libg4c.so from the CURRENT sources: run make again after a git stash pop)
# usage: build_ws_timing.sh [suffix [extra -D flags...]]   (e.g. build_ws_timing.sh _a1 -DG4C_WS_ABLATE=1)
SFX=$1; [ $# -gt 0 ] && shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DG4C_WS_TIMING "$@" -c mlp_ws.hip -o build/mlp_ws_timing$SFX.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libg4c_ws_timing$SFX.so build/error.o build/plan.o build/segment_reduce.o build/mlp_fused.o build/mlp_bx6i.o build/mlp_ws_timing$SFX.o build/mlp_rs.o build/remus_ops.o build/train_ops.o build/knn_grid.o
