#!/bin/bash
# Round-5 evidence, second collection: what the node network's dense trunk, the device-count Adam and the one-wave loss finalisation touch
# (the bench step's kernels are unchanged: profiles/r05_kernel_stats.csv, r05_hbm_traffic.json, r05_tile_kernel_counters.json stay). Writes
# into gpurun_out/r05/ next to the first collection's files; `python tools/collect_round2.py r05` copies them into profiles/.
R=/root/repo; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
(python tools/bench_backend_map.py --eager 2> /dev/null | tail -1; python tools/bench_backend_map.py 2> /dev/null | tail -1) > $O/backend_map.jsonl
python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/mapping_iteration_launches_static.json 2> /dev/null
python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic.json 2> /dev/null
GSR_DENSE_TRUNK=0 GSR_NETWORK_ADAM=0 python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic_library_trunk.json 2> /dev/null
python tools/dev_determinism.py 36 320 240 2>/dev/null | tail -4 > $O/dynamic_reproducibility.txt
python tools/bench_tracking.py 2> /dev/null | tail -1 > $O/tracking_graph.json
(for r in 33280 66560; do python tools/dev_dense.py $r 2> /dev/null | head -1; python tools/dev_dense_chain.py $r 1 2> /dev/null | tail -1; done) > $O/dense_layers.jsonl
python tools/run_slam_demo.py > $O/slam_demo.json 2> $O/slam_demo.err
python tools/run_config4_stand_in.py > $O/config4_stand_in.json 2> /dev/null
ls -la $O | head -40
