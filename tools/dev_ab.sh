cd /root/repo/4dgs-slam_amd
for variant in "" "-fno-slp-vectorize"; do
  ./csrc/build.sh $variant > /dev/null 2>&1
  echo "variant: [$variant]"
  python ../bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.4e ms/step %.3f' % (d['value'], d['ms_per_step'])); print(d['kernel_us'])
"
done
