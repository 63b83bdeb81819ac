#!/bin/bash
# dev: SQ counters of the bf16-split MLP forward kernel (two PMC passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/mlp_pmc; rm -rf $O; mkdir -p $O
export GSR_MLP_RT=${GSR_MLP_RT:-2}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o m -- python $R/tools/dev_mlp_bench.py 2000000 > /dev/null 2> $O/e1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p2 -o m -- python $R/tools/dev_mlp_bench.py 2000000 > /dev/null 2> $O/e2
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'deform_mlp_fwd' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    for n in sorted(m): print('   %-28s %16.0f' % (n, m[n]))
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8
    if cyc:
        print('   kernel cycles %.0f  MfmaUtil %.1f%%  mean waves/SIMD %.2f' % (cyc, 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024), 4 * m.get('SQ_WAVE_CYCLES', 0) / (1024 * cyc)))
PY
tail -3 $O/e1
