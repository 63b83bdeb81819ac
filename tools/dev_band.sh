cd /root/repo/4dgs-slam_amd
for v in 0 1; do
  ./csrc/build.sh -DGSR_NO_BANDING=$v > /dev/null 2>&1
  echo "NO_BANDING $v"
  python ../bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'], {k: d['kernel_us'][k] for k in ('sort_tiles', 'render_fwd','render_bwd')})
"
done
