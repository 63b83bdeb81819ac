#!/bin/bash
# dev: SQ counters of dense_wgrad_many_kernel (two PMC passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wgm_pmc; rm -rf $O; mkdir -p $O
export WGM_WHICH=many
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o m -- python $R/tools/dev_wgrad_many.py 33280 > /dev/null 2> $O/e1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p2 -o m -- python $R/tools/dev_wgrad_many.py 33280 > /dev/null 2> $O/e2
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/p3 -o m -- python $R/tools/dev_wgrad_many.py 33280 > /dev/null 2> $O/e3
python - $O <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'wgrad_many_kernel' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    for n in sorted(m): print('   %-28s %16.0f' % (n, m[n]))
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8
    if cyc:
        print('   kernel cycles %.0f  MfmaUtil %.1f%%  mean waves/SIMD %.2f' % (cyc, 100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024), 4 * m.get('SQ_WAVE_CYCLES', 0) / (1024 * cyc)))
PY
tail -2 $O/e1 $O/e3
find $O -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' \) -delete
