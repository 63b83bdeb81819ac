#!/bin/bash
# dev A/B on ONE box: the tracking iteration with / without the round-6 folds
cd /root/repo
for rep in 1 2; do
echo "--- default";                 python tools/dev_track_probe.py 2>/dev/null | tail -1
echo "--- GSR_RAW_HIST_BLOCKS=0";   GSR_RAW_HIST_BLOCKS=0 python tools/dev_track_probe.py 2>/dev/null | tail -1
echo "--- GSR_TRACK_STEP=0";        GSR_TRACK_STEP=0 python tools/dev_track_probe.py 2>/dev/null | tail -1
done
