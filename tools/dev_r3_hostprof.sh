#!/bin/bash
# Where does a SLAM run spend its wall time? cProfile of the host + rocprofv3 kernel stats of the same run (dev probe).
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r03; mkdir -p $O $O/hostprof
cd $R
GSR_GLUE=ctypes GSR_LIB=$R/4dgs-slam_amd/_timing/libgs_timing.so python tools/phase_cycles.py --json > $O/phase_cycles.json 2> $O/phase_cycles.err
python tools/run_slam_demo.py --only dynamic_320x240_graph --profile > $O/hostprof/dyn.json 2> $O/hostprof/dyn_cprofile.txt
python tools/run_slam_demo.py --only static_640x480_graph --profile > $O/hostprof/static.json 2> $O/hostprof/static_cprofile.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hostprof/dyn_stats -o dyn -- python $R/tools/run_slam_demo.py --only dynamic_320x240_graph > $O/hostprof/dyn_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hostprof/static_stats -o st -- python $R/tools/run_slam_demo.py --only static_640x480_graph > $O/hostprof/static_under_rocprof.json 2> /dev/null
find $O/hostprof -type f \( -name '*kernel_trace.csv' -o -name '*agent_info.csv' -o -name '*.db' -o -name '*.rocpd' \) -delete
ls -la $O/hostprof $O/hostprof/*
