cd /root/repo
for fs in 1 0 1 0; do
GSR_FUSE_SORT=$fs python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']; print('fuse=$fs', ' '.join('%s %.1f' % (n[:10], v) for n, v in k.items()), 'sum %.1f' % sum(k.values()), 'step %.1f us' % (d['ms_per_step'] * 1e3))
"
done
