#!/bin/bash
# dev: the aten ops (with shapes) left in the dynamic mapping iteration
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/aten; rm -rf $O; mkdir -p $O
timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 --aten --eager > $O/launches.json 2> $O/aten.txt
grep " us " $O/aten.txt | head -90
