#!/bin/bash
# dev: long-list workloads (bench.py --workload long) for each variant library under 4dgs-slam_amd/_variants/*.so on ONE box
cd /root/repo
for lib in 4dgs-slam_amd/_variants/*.so; do
  for pt in "0.01 0" "0.015 0" "0.02 0" "0.03 3" "0.05 3"; do
    set -- $pt
    GSR_GLUE=ctypes GSR_LIB=$PWD/$lib python bench.py --workload long --scale-mean $1 --sh-degree $2 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k = d['kernel_us']
print('%-16s scale %-6s R %8d ' % ('$lib'.split('/')[-1], '$1', d['config']['instances']), ' '.join('%s %.1f' % (n[:9], v) for n, v in k.items()), 'step %.1f us' % (d['ms_per_step'] * 1e3))
"
  done
done
