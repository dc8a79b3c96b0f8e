#!/bin/bash
# build_ws_variant.sh <name> [extra -D flags...]: graphs4cfd_amd/lib/libg4c_ws_<name>.so = the shipped library with mlp_ws.hip rebuilt
# under the extra defines (same-box A/Bs of the weight-stationary kernel: scripts/ws_ab_variants.py)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../graphs4cfd_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c mlp_ws.hip -o build/mlp_ws_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libg4c_ws_$NAME.so build/error.o build/plan.o build/segment_reduce.o build/mlp_fused.o build/mlp_bx6i.o build/mlp_ws_$NAME.o build/mlp_rs.o build/remus_ops.o build/train_ops.o build/knn_grid.o
