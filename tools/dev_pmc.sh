cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o r01 -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_stats_bench.json 2> $O/prof_stats.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/prof_pmc1 -o p1 -- $B > /dev/null 2> $O/prof_pmc1.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/prof_pmc2 -o p2 -- $B > /dev/null 2> $O/prof_pmc2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof_pmc3 -o p3 -- $B > /dev/null 2> $O/prof_pmc3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof_pmc4 -o p4 -- $B > /dev/null 2> $O/prof_pmc4.err
find $O -name "*.csv" | head -30
tail -3 $O/prof_pmc1.err
