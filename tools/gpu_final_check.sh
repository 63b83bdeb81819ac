#!/bin/bash
# the round's closing check on a GPU box: the whole -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out
{
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 | cut -c1-600
} > gpurun_out/final_check.txt 2>&1
cat gpurun_out/final_check.txt
