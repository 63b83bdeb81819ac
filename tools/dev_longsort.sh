#!/bin/bash
# dev: sort_tiles time of the long-list workloads for the chunk + rank knobs
cd /root/repo
for cfg in "4096 4096" "1024 1024" "1024 2048" "1024 4096" "4096 2048"; do
  set -- $cfg
  for sm in 0.02 0.05; do
    GSR_LONG_FROM=$1 GSR_LONG_CHUNK=$2 timeout 120 python bench.py --workload long --scale-mean $sm --sh-degree 0 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('from $1 chunk $2 scale $sm: ms', round(d['ms_per_step'],3), 'sort', d['kernel_us'].get('sort_tiles'), 'fwd', d['kernel_us']['render_fwd'])"
  done
done
