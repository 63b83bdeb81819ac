#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/full; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
grep -v "Warning\|warn" $O/tests.log | tail -40
