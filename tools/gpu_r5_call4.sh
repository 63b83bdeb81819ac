#!/bin/bash
# round 5: which speculative capacity do the redone graph runs outgrow? (stand-in of config #4, dynamic census)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5c4; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/run_config4_stand_in.py > $O/config4.json 2> $O/config4.err; tail -2 $O/config4.err
timeout 900 python tools/mapping_iteration_launches.py --wh 640 480 > $O/dyn.json 2> $O/dyn.err; tail -2 $O/dyn.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5c4/config4.json'))
print({k: d[k] for k in ('seconds', 'fps', 'gaussians', 'ate_rmse')})
for k, v in d['mapping_graph_stats'].items():
    print(k, {a: b for a, b in v.items() if a != 'overflow_causes'})
    for c in v.get('overflow_causes', []):
        print('   ', c)
d = json.load(open('gpurun_out/r5c4/dyn.json'))
gs = d.get('graph_stats') or d
print(json.dumps(gs)[:3000])
PY
