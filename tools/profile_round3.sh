#!/bin/bash
# Round-3 evidence under gpurun_out/r03/ (copied into profiles/ by `python tools/collect_round2.py r03`):
#   rocprofv3 kernel stats + HBM traffic (separate PMC passes) of the default bench, SQ counter passes of the tile kernels, the bench line,
#   per-wave phase cycles of every kernel of the step (a -DGSR_FWD_TIMING=1 build of the library), the multi-view entry point vs view-by-view
#   calls, the back-end's mapping iteration, long-list workloads, the SLAM runs.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; if [ -z "$CORE" ]; then rm -rf $O; fi; mkdir -p $O; rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r03 -- $CMD > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r03 -- $CMD > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o r03 -- $CMD > /dev/null 2> $O/pmc_sq.err
cd $R
bash tools/profile_counters.sh r03 > /dev/null 2>&1
timeout 400 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
if [ -f 4dgs-slam_amd/_timing/libgs_timing.so ]; then
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py --json > $O/phase_cycles.json 2> /dev/null
fi
if [ -f 4dgs-slam_amd/_timing/libgs_timing.so ]; then
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py --gaussians 30000 --scale-mean 0.03 --json > $O/phase_cycles_slam_scale.json 2> /dev/null
fi
python tools/bench_views.py 2> /dev/null | tail -1 > $O/views.json
python tools/bench_views.py --dyn 2> /dev/null | tail -1 > $O/views_deltas.json
python tools/bench_views.py --gaussians 100000 --scale-mean 0.01 2> /dev/null | tail -1 > $O/views_100k.json
(GSR_MULTI_VIEW=0 python tools/bench_backend_map.py 2> /dev/null | tail -1; python tools/bench_backend_map.py 2> /dev/null | tail -1) > $O/backend_map.jsonl
python tools/bench_long_lists.py > $O/long_lists.json 2> /dev/null
if [ -z "$CORE" ]; then
python tools/bench_tracking.py 2> /dev/null | tail -1 > $O/tracking_graph.json
python tools/bench_config3.py --fused-only 2> /dev/null | tail -1 > $O/config3.json
python tools/bench_render_wrapper.py 2> /dev/null | tail -1 > $O/render_wrapper.json
for a in "" "--flow" "--nodes" "--nodes --flow"; do python tools/bench_mapping_iteration.py $a 2>/dev/null | tail -1 > "$O/mapping_iteration$(echo $a | tr -d ' -').json"; done
python tools/run_slam_demo.py > $O/slam_demo.json 2> $O/slam_demo.err
python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic.json 2> /dev/null
python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/mapping_iteration_launches_static.json 2> /dev/null
fi
# keep what the collector reads; the traces themselves are bulky
find $O $R/gpurun_out/counters_r03 -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' -o -name '*.rocpd' \) -delete
du -sh $R/gpurun_out
ls $O
