#!/bin/bash
# round 6, GPU call 1: A/B of the first variant batch + the residency timeline of the two tile kernels
cd /root/repo; mkdir -p gpurun_out/r06
REPS=3 bash tools/dev_ab.sh > gpurun_out/r06/ab1.txt 2>&1
GSR_GLUE=ctypes GSR_LIB=$PWD/4dgs-slam_amd/_variants/timeline.so python tools/tile_timeline.py --json > gpurun_out/r06/timeline.json 2> gpurun_out/r06/timeline.err
cat gpurun_out/r06/ab1.txt; cat gpurun_out/r06/timeline.json; tail -3 gpurun_out/r06/timeline.err
