#!/bin/bash
# dev: build the product and the exact-math library; /tmp/gsr_build_done appears when both are finished (removed at the start)
rm -f /tmp/gsr_build_done
cd /root/repo/4dgs-slam_amd/csrc
bash build.sh > /tmp/gsr_build1.log 2>&1; echo "product rc=$?" > /tmp/gsr_build_status
if [ "$1" != "--product" ]; then bash build.sh --exact > /tmp/gsr_build2.log 2>&1; echo "exact rc=$?" >> /tmp/gsr_build_status; fi
touch /tmp/gsr_build_done
