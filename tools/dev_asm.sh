#!/bin/bash
# dev: device assembly of gs_capi.hip -> /tmp/gs.s, one kernel cut out to /tmp/<name>.s  (usage: dev_asm.sh render_fwd [extra hipcc flags])
K=${1:-render_fwd}; shift
cd /root/repo/4dgs-slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-inline-asm -fno-slp-vectorize --cuda-device-only -S gs_capi.hip -o /tmp/gs.s "$@" 2>&1 | grep -v "hip-link"
awk -v k="$K" '$0 ~ "^_ZN3gsr[0-9]+" k "_kernel" {on=1} on {print} on && /^\.Lfunc_end/ {exit}' /tmp/gs.s > /tmp/$K.s
grep -n "NumVgprs:\|Occupancy:\|LDSByteSize\|ScratchSize" /tmp/$K.s
wc -l /tmp/$K.s
