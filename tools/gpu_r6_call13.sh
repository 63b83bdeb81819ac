#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
REPS=3 bash tools/dev_ab.sh > gpurun_out/r06/ab13.txt 2>&1
cat gpurun_out/r06/ab13.txt

