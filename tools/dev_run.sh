cd /root/repo
python tests/dev/dev_parity.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); worst = max(v for k, v in d.items() if isinstance(v, float)); print(d['tag'], 'worst %.2e' % worst, {k: v for k, v in d.items() if 'mismatch' in k and v})
    except Exception: print(l.rstrip())
"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.4e ms/step %.3f' % (d['value'], d['ms_per_step'])); print(d['kernel_us'])
"
