cd /root/repo/4dgs-slam_amd
for gb in 1024 512 256; do
  sed -i "s/^constexpr int GB = [0-9]*;/constexpr int GB = $gb;/" csrc/gs_forward.h
  ./csrc/build.sh > /dev/null 2>&1
  echo "GB $gb"
  python ../bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'], {k: d['kernel_us'][k] for k in ('preprocess_fwd','scan','scatter_instances','sort_tiles')})
"
done
