#!/bin/bash
# dev (round 3): parity tests of the default library, then A/B per-kernel times of every library under _variants/
cd /root/repo
if [ "$1" != "notest" ]; then timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5; fi
bash tools/dev_ab.sh
