#!/bin/bash
# the cycle-accounting build of the library for tools/phase_cycles.py (never loaded by the product: pass it through GSR_LIB)
mkdir -p /root/repo/4dgs-slam_amd/_timing
cd /root/repo/4dgs-slam_amd/csrc && ./build.sh -DGSR_FWD_TIMING=1 -o ../_timing/libgs_timing.so
