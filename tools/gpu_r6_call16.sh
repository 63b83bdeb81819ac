cd /root/repo
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_slam.py tests/test_hip_bindings.py tests/test_hip_fused_prologue.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v Warning | tail -8
bash tools/gpu_r6_tracking.sh 2>&1 | tail -14
python tools/dev_track_probe.py 2>/dev/null | tail -1
