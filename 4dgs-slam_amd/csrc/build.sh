#!/bin/bash
# Builds libgs_rasterizer_hip.so (the C-ABI library of include/gs_rasterizer.h + include/simple_knn.h) for gfx950, in-tree.
# -fno-slp-vectorize: hipcc otherwise packs scalar f32 math into v_pk_* with extra v_mov shuffles (render_bwd: 267 -> 249 us).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize gs_capi.hip -o ../libgs_rasterizer_hip.so "$@"
echo "built $(cd .. && pwd)/libgs_rasterizer_hip.so"
