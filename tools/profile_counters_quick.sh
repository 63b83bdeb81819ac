#!/bin/bash
# two SQ passes only (occupancy / wait split) -> gpurun_out/counters_<tag>/
TAG=${1:-q}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/counters_$TAG
rm -rf $O && mkdir -p $O
B="python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/sq1 -o p -- $B > /dev/null 2> $O/sq1.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/sq3 -o p -- $B > /dev/null 2> $O/sq3.err
