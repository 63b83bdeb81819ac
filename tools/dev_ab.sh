#!/bin/bash
# A/B kernel timing on ONE box: runs the bench's per-kernel timing for each variant library under 4dgs-slam_amd/_variants/*.so
# (built locally with: cd 4dgs-slam_amd/csrc && ./build.sh -DGSR_VARIANT_X=1 -o ../_variants/x.so), twice, interleaved.
cd /root/repo
for rep in 1 2; do
  for lib in 4dgs-slam_amd/_variants/*.so; do
    GSR_GLUE=ctypes GSR_LIB=$PWD/$lib python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']; print('%-28s' % '$lib'.split('/')[-1], ' '.join('%s %.1f' % (n[:8], v) for n, v in k.items()), 'sum %.1f' % sum(k.values()), 'step %.1f us' % (d['ms_per_step'] * 1e3))
"
  done
done
