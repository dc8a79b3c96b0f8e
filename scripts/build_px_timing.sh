#!/bin/bash
# libg4c variant whose persistent MLP kernel records phase stamps (scripts/px6_stamps.py): build_px_timing.sh <out.so> [-D...]
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../graphs4cfd_amd/csrc"
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DG4C_PX_TIMING "$@" -c mlp_px6.hip -o build/mlp_px6_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" build/error.o build/plan.o build/segment_reduce.o build/mlp_fused.o build/mlp_px6_timing.o build/mlp_bx6i.o build/remus_ops.o build/train_ops.o
