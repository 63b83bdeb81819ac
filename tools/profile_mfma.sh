#!/bin/bash
# MFMA utilisation of the fused deformation-MLP kernels (PMC pass only: --kernel-trace + --pmc, nothing else).
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/mfma_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o m -- python /root/repo/tools/bench_deformation.py --n 200000 --iters 5 --only-network > $O/out.json 2> $O/err
python - $O <<'PY'
import csv, glob, collections, sys, json
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'gsr::deform_mlp' in k or 'linear_wgrad_kernel' in k:
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in d.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    # MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100, SIMD_NUM = 1024 on MI355X
    m['MfmaUtil_percent'] = 100.0 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024)   # counters are summed over the 8 XCDs
    m['mfma_flops'] = m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512
    out[k] = m
    print(k, {n: round(v, 1) for n, v in m.items()})
json.dump(out, open(sys.argv[1] + '/mfma.json', 'w'), indent=1)
PY
