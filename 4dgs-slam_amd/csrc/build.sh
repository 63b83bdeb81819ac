#!/bin/bash
# Builds libgs_rasterizer_hip.so (the C-ABI library of include/*.h) for gfx950, in-tree.
#   build.sh            the product library
#   build.sh --exact    libgs_rasterizer_hip_exact.so: the exact-math parity variant (GSR_EXACT_MATH=1 selects it at import time):
#                       expf-equivalent exponential, true division, reference operation order in the tile kernels, no fp contraction
# -fno-slp-vectorize: hipcc otherwise packs scalar f32 math into v_pk_* with extra v_mov shuffles (render_bwd: 267 -> 249 us).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libgs_rasterizer_hip.so
EXTRA=()
if [ "$1" = "--exact" ]; then
  shift
  OUT=../libgs_rasterizer_hip_exact.so
  EXTRA=(-DGSR_EXACT_MATH=1 -ffp-contract=off)
fi
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize "${EXTRA[@]}" gs_capi.hip -o $OUT "$@"
echo "built $(cd .. && pwd)/$(basename $OUT)"
