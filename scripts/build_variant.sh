#!/bin/bash
# build_variant.sh <out.so> [extra hipcc -D flags...]: libg4c variant with extra defines for mlp_fused.hip (A/B tuning)
# G4C_SRC=<file in csrc/> builds the variant from another copy of the kernel source
set -e
OUT=$1; shift
cd "$(dirname "$0")/../graphs4cfd_amd/csrc"
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c "${G4C_SRC:-mlp_fused.hip}" -o build/mlp_fused_variant.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" build/error.o build/plan.o build/segment_reduce.o build/mlp_fused_variant.o build/mlp_bx6i.o build/mlp_ws.o build/mlp_rs.o build/remus_ops.o build/train_ops.o build/knn_grid.o
