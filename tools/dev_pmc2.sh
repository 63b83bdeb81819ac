cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
O=/root/repo/gpurun_out
rm -rf $O/pm1 $O/pm2 $O/pm3
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pm1 -o p -- $B > /dev/null 2> $O/pm1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/pm2 -o p -- $B > /dev/null 2> $O/pm2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES_RESTORED SQ_INSTS_VALU_TRANS SQ_INSTS_WAVE32_LDS --kernel-trace --output-format csv -d $O/pm3 -o p -- $B > /dev/null 2> $O/pm3.err
tail -2 $O/pm3.err
