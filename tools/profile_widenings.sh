#!/bin/bash
# rocprofv3 artefacts for the SURVEY 8(f) rank-3 widenings (deformation field + control nodes), committed under profiles/
# by tools/collect_widening_profiles.py.  Run on the GPU box via gpurun.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/widen_$TAG
rm -rf $O; mkdir -p $O
DEF="python /root/repo/tools/bench_deformation.py --n 200000 --iters 10 --only-network"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/def_stats -o def -- $DEF > $O/def_under_rocprof.json 2> $O/def.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/def_fetch -o def -- $DEF > /dev/null 2>> $O/def.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/def_write -o def -- $DEF > /dev/null 2>> $O/def.err
NOD="python /root/repo/tools/bench_control_nodes.py --n 100000 --iters 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nod_stats -o nod -- $NOD > $O/nod_under_rocprof.json 2> $O/nod.err
python /root/repo/tools/bench_deformation.py --n 200000 > $O/deformation_200k.json 2>> $O/def.err
python /root/repo/tools/bench_deformation.py --n 500000 --iters 10 > $O/deformation_500k.json 2>> $O/def.err
python /root/repo/tools/bench_control_nodes.py --n 100000 > $O/control_nodes_100k.json 2>> $O/nod.err
python /root/repo/tools/bench_control_nodes.py --n 20000 > $O/control_nodes_20k.json 2>> $O/nod.err
ls $O
