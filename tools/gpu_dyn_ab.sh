#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_control_nodes.py -x -q -k "relu or fused_network" 2>&1 | tail -2
for cfg in "1 2000,7000" "0 2000,7000"; do
  set -- $cfg
  GSR_FUSED_TRUNK=$1 GSR_DYN_MARGINS=$2 timeout 600 python tools/mapping_iteration_launches.py --wh 640 480 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
g=d['graph']
print('trunk=$1 margins=$2', 'graph ms/it %.3f (incl capture %.3f)' % (g['ms_per_iteration_without_capture'], g['ms_per_iteration_incl_capture']), 'direct ms/it %.3f launches %.0f device us %.0f' % (d['ms_per_iteration'], d['launches_per_iteration'], d['device_us_per_iteration']), g['second_call'])
k=d['device_us_per_iteration_by_kernel']
for a,b in list(k.items())[:30]: print('   %8.1f %s' % (b, a))
"
done
