#!/bin/bash
# Round-6 evidence under gpurun_out/r06/ (copied into profiles/ by `python tools/collect_round2.py r06`):
#   rocprofv3 kernel stats + HBM traffic (separate PMC passes) of the default bench, SQ counter passes of the tile kernels, the bench line,
#   per-wave phase cycles (a -DGSR_FWD_TIMING=1 build), BackEnd.map_static() eager vs hipGraph replays, the launch censuses of the mapping
#   iterations, the SLAM runs, the reproducibility probe of the dynamic branch, the N = 2 code path of bench.py on one GPU (gloo), BASELINE
#   config #3 (reference program / per-view / batched keyframes) with the batched iteration's rocprofv3 kernel stats.
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06; if [ -z "$CORE" ]; then rm -rf $O; fi; mkdir -p $O; rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r06 -- $CMD > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r06 -- $CMD > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r06 -- $CMD > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o r06 -- $CMD > /dev/null 2> $O/pmc_sq.err
cd $R
bash tools/profile_counters.sh r06 > /dev/null 2>&1
timeout 500 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
if [ -f 4dgs-slam_amd/_timing/libgs_timing.so ]; then
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py --json > $O/phase_cycles.json 2> /dev/null
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py --gaussians 30000 --scale-mean 0.03 --json > $O/phase_cycles_slam_scale.json 2> /dev/null
fi
(python tools/bench_backend_map.py --eager 2> /dev/null | tail -1; python tools/bench_backend_map.py 2> /dev/null | tail -1) > $O/backend_map.jsonl
python tools/mapping_iteration_launches.py --static --wh 640 480 > $O/mapping_iteration_launches_static.json 2> /dev/null
python tools/mapping_iteration_launches.py --wh 640 480 > $O/mapping_iteration_launches_dynamic.json 2> /dev/null
python tools/bench_long_lists.py > $O/long_lists.json 2> /dev/null
python tools/dev_determinism.py 36 320 240 2>/dev/null | tail -4 > $O/dynamic_reproducibility.txt
if [ -f 4dgs-slam_amd/_variants/timeline.so ]; then   # residency timeline of the two tile kernels + this box's dispatch pattern (-DGSR_TIMELINE=1 build)
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_variants/timeline.so python tools/tile_timeline.py --json > $O/tile_timeline.json 2> /dev/null
  GSR_ORDER_ITEMS=0 GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_variants/timeline.so python tools/tile_timeline.py --json > $O/tile_timeline_tile_order.json 2> /dev/null
  GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_variants/timeline.so python tools/dev_dispatch_census.py > $O/dispatch_census.json 2> /dev/null
fi
GSR_BENCH_DEVICE=0 GSR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_two_ranks_one_gpu_gloo.json 2> /dev/null
if [ -z "$CORE" ]; then
python tools/bench_views.py 2> /dev/null | tail -1 > $O/views.json
python tools/bench_tracking.py 2> /dev/null | tail -1 > $O/tracking_graph.json
python tools/dev_track_probe.py 2> /dev/null | tail -1 > $O/tracking_probe.json
GSR_TRACK_STEP=0 python tools/dev_track_probe.py 2> /dev/null | tail -1 > $O/tracking_probe_autograd_route.json
( for e in "GSR_MLP_FP32=1" "GSR_MLP_RT=2" "GSR_MLP_RT=4"; do env $e python tools/dev_mlp_bench.py 2> /dev/null | tail -1; done ) > $O/deform_mlp.jsonl
python tools/bench_config3.py 2> /dev/null | tail -1 > $O/config3.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/config3_stats -o r06 -- python $R/tools/bench_config3.py --modes batched --iters 3 > /dev/null 2> $O/config3_stats.err )
cp $(find $O/config3_stats -name '*kernel_stats.csv' | head -1) $O/config3_kernel_stats.csv 2> /dev/null; rm -rf $O/config3_stats
python tools/run_slam_demo.py > $O/slam_demo.json 2> $O/slam_demo.err
python tools/run_config4_stand_in.py > $O/config4_stand_in.json 2> /dev/null
fi
find $O $R/gpurun_out/counters_r06 -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' -o -name '*.rocpd' \) -delete
du -sh $R/gpurun_out
ls $O
