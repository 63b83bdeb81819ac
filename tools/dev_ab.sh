#!/bin/bash
# A/B kernel timing on ONE box: runs the bench's per-kernel timing for each variant library under 4dgs-slam_amd/_variants/*.so
# (built locally with tools/dev_build_variants.sh name:"-DGSR_VARIANT_X=1" ...), REPS times (default 2), interleaved.
# A variant may come with environment settings: <name>.env beside <name>.so (one line, VAR=value ...).
# The timeline build (timeline.so) is skipped: it is read by tools/tile_timeline.py.
cd /root/repo
# which dispatch pattern this box has (the forward pass's tile dealing assumes round robin over 32 CUs per XCD): one line, when the timeline build is there
[ -f 4dgs-slam_amd/_variants/timeline.so ] && GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_variants/timeline.so python tools/dev_dispatch_census.py 2>/dev/null | tail -1
for rep in $(seq 1 ${REPS:-2}); do
  for lib in 4dgs-slam_amd/_variants/*.so; do
    case $lib in *timeline*|*timing*) continue;; esac
    env $( [ -f ${lib%.so}.env ] && cat ${lib%.so}.env ) GSR_GLUE=ctypes GSR_LIB=$PWD/$lib python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-secondary --graph-replay 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']; print('%-28s' % '$lib'.split('/')[-1], ' '.join('%s %.1f' % (n[:8], v) for n, v in k.items()), 'sum %.1f' % sum(k.values()), 'step %.1f us' % (d['ms_per_step'] * 1e3), 'graph %.1f' % ((d.get('graph_replay') or {}).get('ms_per_step', 0) * 1e3))
"
  done
done
