cd /root/repo
for pb in 0 512 640 768 896 1024; do
  echo "PERSIST $pb"
  GSR_PERSIST_BLOCKS=$pb python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.3f' % d['ms_per_step'], {k: d['kernel_us'][k] for k in ('render_fwd','render_bwd')})
"
done
