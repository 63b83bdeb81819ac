#!/bin/bash
# Produces the rocprofv3 artefacts committed under profiles/ (run on the GPU box via gpurun):
#   kernel-trace --stats of the default bench command, and separate --pmc passes for HBM traffic (FETCH_SIZE, WRITE_SIZE).
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/profile_$TAG
mkdir -p $O
CMD="python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o $TAG -- $CMD > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o $TAG -- $CMD > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o $TAG -- $CMD > /dev/null 2> $O/pmc_sq.err
python /root/repo/bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
ls $O $O/stats
