#!/bin/bash
mkdir -p gpurun_out
{
python -m pytest tests/test_hip_dense.py -x -q 2>&1 | tail -8
for r in 32768 33280 66560; do
  for w in 0 1; do echo "rows $r wide $w"; GSR_DENSE_WIDE=$w python tools/dev_dense.py $r 2>/dev/null | head -1; done
done
} > gpurun_out/dense_tests.txt 2>&1
cat gpurun_out/dense_tests.txt
