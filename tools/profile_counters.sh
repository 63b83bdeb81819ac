#!/bin/bash
# SQ / GRBM counter passes (and the derived metrics rocprofv3 offers) for the two tile kernels -> gpurun_out/counters_<tag>/;
# tools/collect_counters.py turns them into profiles/<tag>_tile_kernel_counters.json. PMC passes only carry --kernel-trace.
TAG=${1:-r02}
EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/counters_$TAG
rm -rf $O && mkdir -p $O
B="python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $EXTRA"
pass() {  # name counters...
  local n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- $B > /dev/null 2> $O/$n.err || echo "pass $n failed: $(tail -1 $O/$n.err)"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass sq3 GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
pass sq4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES
pass d1 VALUBusy VALUUtilization SALUBusy
pass d2 LDSBankConflict MemUnitStalled OccupancyPercent
pass d3 MemUnitBusy L2CacheHit
ls $O
